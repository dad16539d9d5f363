import sys, pathlib
src = (pathlib.Path(__file__).parent / "microbench.py").read_text().split('if __name__ == "__main__":')[0]
exec(compile(src, "microbench", "exec"))
REPS = 2
flash(40, 4096, 128)
