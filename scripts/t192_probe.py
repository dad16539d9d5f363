#!/usr/bin/env python
"""The pipelined loop on 256 x 192 tiles (gemm16_tile = 8) against the per-shape choice on the DiT Linears whose 256 x 256 tile counts quantise badly on 256 CUs."""
import sys

import t320_check as T

TILES = ((-1, 1), (8, 1))
cases = ((4096, 3072, 3072), (4096, 12288, 3072), (4096, 3072, 9216), (4352, 15360, 3072), (4352, 3072, 21504), (4096, 3072, 12288)) if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
ok = True
for rows, K, M in cases:
    ok &= T.linear(rows, K, M, tiles=TILES)
print("ALL OK" if ok else "MISMATCH")
