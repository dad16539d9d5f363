#!/usr/bin/env python
"""Which pipeline bounds the 256x160 implicit-GEMM conv loop?  Times the SD1.5 level-0 3x3 conv (batch 16, 320 -> 320 @ 64x64, the shape
behind bench.py's dominant kernel) four ways through the C ABI: the product kernel, the explicit LDS-read / MFMA interleave
(gemm16_sched = 1), and two ABLATIONS that compute wrong results on purpose — DMA stream + barriers only (16), fragment reads + MFMAs
only (32).  Run under `rocprofv3 --kernel-trace --stats` for exact kernel durations; the wall numbers printed here include packing."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import microbench as mb  # noqa: E402  (loads the backend)

for opt, label in ((0, "product kernel"), (1, "explicit LDS-read / MFMA interleave"), (16, "ablation: DMA + barriers only"), (32, "ablation: reads + MFMAs only")):
    mb.blib.ggml_backend_mi355x_set_option(b"gemm16_sched", opt)
    print(f"--- gemm16_sched = {opt}: {label}")
    mb.conv(16, 320, 320, 64)
    mb.conv(16, 640, 640, 32)
mb.blib.ggml_backend_mi355x_set_option(b"gemm16_sched", 0)
