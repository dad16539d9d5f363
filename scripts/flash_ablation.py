#!/usr/bin/env python
"""Where do the 380 us of the d = 40 self-attention launch go?  Times flash(d 40, L 4096, 128 head-images) with the product kernel and
two ABLATIONS that compute wrong results on purpose: no softmax VALU work (1), no K/V restaging (2).  Run under rocprofv3
--kernel-trace --stats for exact kernel durations."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import microbench as mb  # noqa: E402

for opt, label in ((0, "product kernel"), (1, "ablation: no softmax VALU"), (2, "ablation: no K/V restaging")):
    mb.blib.ggml_backend_mi355x_set_option(b"flash_ablate", opt)
    print(f"--- flash_ablate = {opt}: {label}")
    mb.flash(40, 4096, 128)
mb.blib.ggml_backend_mi355x_set_option(b"flash_ablate", 0)
