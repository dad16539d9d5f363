#!/usr/bin/env python
"""Does running the batch as independent image groups on SEPARATE streams overlap the memory-bound kernels of one group with the MFMA-bound kernels of
the other?  One engine x 16 UNet rows (8 images x cond / uncond) against G engines (own backend instance = own stream, own weight copy) x 16 / G rows each,
driven by G host threads.  usage: two_stream_probe.py [G ...]"""
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import sdcpp_amd as sd

sd.load_mi355x_backend()
rng = np.random.default_rng(0)
STEPS = 6


def inputs(rows):
    return (rng.standard_normal((rows, 4, 64, 64)).astype(np.float32), np.full((rows,), 500.0, dtype=np.float32),
            rng.standard_normal((2, 77, 768)).astype(np.float32))


def run(engs, ins, steps):
    bar = threading.Barrier(len(engs) + 1)

    def work(e, i):
        x, t, c = i
        bar.wait()
        for _ in range(steps):
            e.unet_forward(x, t, c, None)
        bar.wait()

    th = [threading.Thread(target=work, args=(e, i)) for e, i in zip(engs, ins)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    return dt / steps * 1e3


for G in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    rows = 16 // G
    engs = [sd.Engine(model=sd.SD15, backend="MI355X0", wtype=sd.F16, flash_attn=True) for _ in range(G)]
    ins = [inputs(rows) for _ in range(G)]
    for e, (x, t, c) in zip(engs, ins):
        e.unet_forward(x, t, c, None)
        e.unet_forward(x, t, c, None)
    res = [run(engs, ins, STEPS) for _ in range(3)]
    one = run(engs[:1], ins[:1], STEPS)
    print(f"{G} stream(s) x {rows} rows: {min(res):7.2f} ms per forward of all 16 rows (runs: {', '.join(f'{r:.2f}' for r in res)});  one group alone: {one:7.2f} ms", flush=True)
    for e in engs:
        e.close()
