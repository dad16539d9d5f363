#!/usr/bin/env python
"""Every gemm16 tile configuration forced in turn on the short-K d -> d GEMMs of the SD1.5 transformer blocks (q/k/v/out projections, proj_out
1x1 convs): which tile the per-shape choice should take.  Columns = gemm16_tile 0..6 (T128, T256, T256W, T160, T160N, T320, T256P)."""
import t320_check as T

TILES = tuple((t, 1) for t in range(7))
ok = True
for rows, d in ((65536, 320), (16384, 640), (4096, 1280)):
    ok &= T.linear(rows, d, d, tiles=TILES)
    ok &= T.linear(rows, d, d, res=True, tiles=TILES)
    hw = {65536: 64, 16384: 32, 4096: 16}[rows]
    ok &= T.conv(16, d, d, hw, ks=1, res=True, tiles=TILES)
print("ALL OK" if ok else "MISMATCH")
