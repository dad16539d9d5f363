#!/usr/bin/env python
"""The pipelined loop on 256 x 192 tiles (gemm16_t192p = 1: chosen per shape) against the choice without it, on the DiT Linears whose 256 x 256 tile counts quantise badly on 256 CUs.
usage: t192_probe.py [rowsxKxM ...]"""
import sys

import t320_check as T
import sdcpp_amd as sd

cases = ((4096, 3072, 3072), (4096, 12288, 3072), (4096, 3072, 9216), (4352, 15360, 3072), (4352, 3072, 21504), (4096, 3072, 12288),
         (8500, 2432, 2432), (8500, 9728, 2432), (8500, 2432, 9728), (8500, 2432, 7296)) if len(sys.argv) < 2 else [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
ok = True
for rows, K, M in cases:
    outs = []
    for v in (0, 1):
        sd.backend_set_option("gemm16_t192p", v)
        print(f"gemm16_t192p={v}: ", end="")
        ok &= T.linear(rows, K, M, tiles=((-1, 1),))
sd.backend_set_option("gemm16_t192p", 1)
print("ALL OK" if ok else "MISMATCH")
