#!/usr/bin/env python
"""In-launch split-K combine on the mid-size Linears: per-shape time with the combine off / on at several workgroup targets."""
import sys

import t320_check as T

sd = T.sd
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(2048, 1280, 1280), (4096, 1280, 1280), (2048, 5120, 1280), (1232, 768, 768), (2048, 640, 640)]
for rows, K, M in shapes:
    for ink, tgt in ((0, 640), (1, 320), (1, 640), (1, 1280)):
        sd.backend_set_option("splitk_inkernel", ink)
        sd.backend_set_option("splitk_in_target", tgt)
        print(f"inkernel={ink} target={tgt}: ", end="")
        T.linear(rows, K, M, res=True, tiles=((-1, 1),), )
sd.backend_set_option("splitk_inkernel", 1)
sd.backend_set_option("splitk_in_target", 640)
