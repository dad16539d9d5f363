"""Generates tests/golden/models_torch_fp32.npz: seeded inputs and PyTorch-CPU fp32 outputs of the tiny-width model graphs (SD1.5 / SDXL
UNet, KL-VAE decoder, SD3.5 MMDiT, FLUX) computed by oracle/torch_ref.py from the engine's synthetic weights (weight seed 1234 — the
weights are a pure function of (seed, tensor name), so the fixture pins oracle AND graph builders without storing the weights).

    python tests/golden/make_model_golden.py     # needs the built host library + CPU oracle (python -c "import __graft_entry__ as g; g.build()")
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import sdcpp_amd as sd  # noqa: E402
from oracle import torch_ref  # noqa: E402

OUT = Path(__file__).resolve().parent / "models_torch_fp32.npz"


def main():
    sd.load_backend(ROOT / "oracle" / "_build" / "libggml-cpu-oracle.so")
    rng = np.random.default_rng(4321)
    d = {}
    # UNets
    for name, ydim in (("SD15_TINY", 0), ("SDXL_TINY", 96)):
        e = sd.Engine(model=getattr(sd, name), backend="CPU-oracle")
        x = rng.standard_normal((2, 4, 16, 16)).astype(np.float32)
        t = np.array([612.0, 45.5], dtype=np.float32)
        ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
        y = rng.standard_normal((1, ydim)).astype(np.float32) if ydim else None
        d[f"{name}_x"], d[f"{name}_t"], d[f"{name}_ctx"] = x, t, ctx
        if y is not None:
            d[f"{name}_y"] = y
        d[f"{name}_out"] = torch_ref.unet_forward(e, name, x, t, ctx, y)
        if name == "SD15_TINY":
            z = (rng.standard_normal((1, 4, 8, 8)) * 0.5).astype(np.float32)
            d["VAE_TINY_z"], d["VAE_TINY_rgb"] = z, torch_ref.vae_decode(e, z)
    # DiTs
    e = sd.Engine(model=sd.SD35_TINY, backend="CPU-oracle")
    x = rng.standard_normal((2, 16, 9, 10)).astype(np.float32)
    t = np.array([820.0, 133.0], dtype=np.float32)
    ctx = rng.standard_normal((2, 18, 96)).astype(np.float32)
    y = rng.standard_normal((2, 64)).astype(np.float32)
    d.update(SD35_TINY_x=x, SD35_TINY_t=t, SD35_TINY_ctx=ctx, SD35_TINY_y=y, SD35_TINY_out=torch_ref.mmdit_forward(e, "SD35_TINY", x, t, ctx, y))
    e = sd.Engine(model=sd.FLUX_TINY, backend="CPU-oracle")
    t = np.array([0.93, 0.21], dtype=np.float32)
    d.update(FLUX_TINY_x=x, FLUX_TINY_t=t, FLUX_TINY_ctx=ctx, FLUX_TINY_y=y, FLUX_TINY_out=torch_ref.flux_forward(e, "FLUX_TINY", x, t, ctx, y))
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, {k: v.shape for k, v in d.items() if k.endswith("out") or k.endswith("rgb")})


if __name__ == "__main__":
    main()
