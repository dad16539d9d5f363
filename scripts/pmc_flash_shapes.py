"""Flash-attention launches of the benchmarked configurations for SQ-level PMC passes: usage pmc_flash_shapes.py d L HN   (SD1.5 64x64 level: 40 4096 128; SDXL: 64 4096 20;
FLUX: 128 4352 24; SD3.5: 64 4250 76)"""
import sys, pathlib
args = [int(a) for a in sys.argv[1:4]]
sys.argv = [sys.argv[0]]
src = (pathlib.Path(__file__).parent / "microbench.py").read_text().split('if __name__ == "__main__":')[0]
exec(compile(src, "microbench", "exec"))
REPS = 2
flash(*args)
