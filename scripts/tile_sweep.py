#!/usr/bin/env python
"""Every gemm16 tile configuration forced in turn on the short-K d -> d GEMMs of the SD1.5 / SDXL transformer blocks (q/k/v/out projections,
proj_out 1x1 convs): which tile the per-shape choice should take.  Columns = gemm16_tile: 0 T128, 1 T256, 4 T160N, 7 128x64
(2 T256W, 3 T160, 5 T320, 6 T256P: pass TILES yourself)."""
import sys

import t320_check as T

TILES = tuple((t, 1) for t in (0, 1, 4, 7))
ok = True
cases = ((65536, 320), (16384, 640), (4096, 1280), (2048, 1280), (8192, 640)) if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for rows, d in cases:
    ok &= T.linear(rows, d, d, tiles=TILES)
    ok &= T.linear(rows, d, d, res=True, tiles=TILES)
print("ALL OK" if ok else "MISMATCH")
