import itertools, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch
junk = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(16)]   # 16 GiB of NaN, then freed: uninitialised reads show up
torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
import sdcpp_amd as sd
sd.load_mi355x_backend()
sd.load_backend(ROOT / "oracle/_build/libggml-cpu-oracle.so")
rng = np.random.default_rng(17)
x = rng.standard_normal((2, 16, 18, 15)).astype(np.float32)
t = np.array([731.0, 210.0], dtype=np.float32)
ctx = rng.standard_normal((1, 154, 96)).astype(np.float32)
y = rng.standard_normal((1, 64)).astype(np.float32)
ref = sd.Engine(model=sd.SD35_TINY, backend="CPU-oracle", flash_attn=False).unet_forward(x, t, ctx, y)
for m, g, ge in itertools.product((0, 1), repeat=3):
    sd.backend_set_option("fuse_modulate", m); sd.backend_set_option("fuse_gate", g); sd.backend_set_option("fuse_gelu", ge)
    out = sd.Engine(model=sd.SD35_TINY, backend="MI355X0", flash_attn=False).unet_forward(x, t, ctx, y)
    print(f"modulate={m} gate={g} gelu={ge}: rel-L2 {np.linalg.norm(out-ref)/np.linalg.norm(ref):.3e}  nan={int(np.isnan(out).sum())}", flush=True)
