import sys
sys.argv = [sys.argv[0]]
import importlib.util, pathlib
spec = importlib.util.spec_from_file_location("mb", pathlib.Path(__file__).parent / "microbench.py")
mb = importlib.util.module_from_spec(spec)
import builtins
# run only two shapes per variant
src = (pathlib.Path(__file__).parent / "microbench.py").read_text().split('if __name__ == "__main__":')[0]
exec(compile(src, "microbench", "exec"))
for v in (0, 1, 3):
    blib.ggml_backend_mi355x_set_option(b"gemm16_variant", v)
    print("--- variant", v)
    conv(16, 640, 640, 32)
    linear(16384, 640, 5120)
flash(40, 4096, 128)
