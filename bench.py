#!/usr/bin/env python
"""bench.py — denoise-loop throughput of the MI355X engine on BASELINE.json's metric.

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a launcher around it: the script re-executes itself as N ranks under torch.distributed.run (one process per GPU, RCCL); launched by
  `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it is one of those ranks.  WORLD_SIZE != N is an error, and `n_gpus` in
  the line is the number of ranks that took part in the closing all-reduce, not the number asked for.

A "step" is one sampler iteration of the hot path over one batch of synthetic input.  Default (device-resident trajectory, SURVEY.md
section 8 f4): the latents stay in HBM for the whole timed region and every step is ONE graph — x*c_in, the cond AND uncond UNet forwards
of every image of the per-GPU batch (cfg 7), the CFG combine and the Euler-A update — exactly what the reference's progress meter counts
as one "it" (src/stable-diffusion.cpp:2470-2482); per step only 8 scalars and the ancestral noise cross PCIe.  With --host-loop a step
is the cond+uncond forward pair driven from host buffers (x up, eps down each step: the reference's boundary; the CFG / Euler host math
is then NOT in the step) — that PCIe-inclusive rate is reported beside the headline as `host_loop`, never as `value`.
Workload at N = 1: BASELINE.json configs[1] (SD1.5 UNet, 512x512, f16, batch 8 on one MI355X).  value = image-iterations per second
over the whole job (batch * steps / time, summed over ranks; max time over ranks).  Images shard across ranks with no data-path
collective => weak scaling: per-GPU batch fixed.

The JSON line also carries
  roofline:     headline = the dominant kernel family (implicit-GEMM conv, 256-row tiles): achieved = sum of the launches' algorithmic
                FLOPs (2 * output positions * IC*KH*KW * OC) / sum of their durations, measured live with HIP events recorded on the
                backend's launch stream around every dispatch, vs the dense f16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).  The timed
                region replays the step's hipGraph (the product default since round 5); events inside a captured graph cannot be read back,
                so the SAME K steps run once more eagerly right after it with the events on (roofline.timing, eager_ms_per_step);
                --hip-graph 0 times eager launches and records the events inside the timed region itself.
                traffic = HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE doubled per the guide's
                gfx950 correction + WRITE_SIZE) with traffic_commit = the version of that file, or null: never measured in this process.
                roofline.kernels[] = EVERY kernel family of the step, timed the same way in a short extra pass after the timed region
                (events around ~800 launches per step would perturb the headline): contraction families against the MFMA peak,
                layout / norm / elementwise families as algorithmic bytes (one read + one write of the activation) / time against
                8 TB/s, each with its share of the step's kernel time.
  sdxl / flux / sd35: the other workloads BASELINE.json names (configs 3, 4, 5) on this GPU's share of each configuration, a few device-resident
                sampler steps after the timed region: ms/step, it/s, whole-step fraction of the MFMA peak, the three heaviest kernel
                families; sdxl's sec_per_image is one TIMED sdm_generate_image call (30 steps + 1024x1024 VAE decode + u8 pixels).
  sdxl_b8:      config 3 at its one-GPU point: the whole batch of 8 SDXL images on this GPU (3 timed steps + one timed generate_image).
  cpu_baseline: the CPU oracle (restatement of the reference ggml-cpu path) timed on this box's host cores on a bounded
                sample of the same workload (rank 0, N = 1 only), with the oracle's default team (<= 16 threads) and with every
                schedulable CPU; value = the faster.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np


class SclkSampler:
    """Shader clock and socket power of GPU 0 while a region runs: a thread asks librocm_smi64 (what `rocm-smi --showclocks --showpower` prints) every 50 ms —
    rsmi_dev_gpu_clk_freq_get(RSMI_CLK_TYPE_SYS): frequency[current], rsmi_dev_current_socket_power_get.  The timed region is a C call that releases the GIL.
    Reports nothing where the library or the calls are missing."""

    class _Freqs(C.Structure):
        _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32), ("frequency", C.c_uint64 * 33)]

    def __init__(self):
        self.mhz, self.watts, self._stop, self._t, self.lib = [], [], False, None, None
        try:
            lib = C.CDLL("librocm_smi64.so")
            if lib.rsmi_init(C.c_uint64(0)) == 0:
                self.lib = lib
        except OSError:
            pass

    def _run(self):
        f, pw = self._Freqs(), C.c_uint64(0)
        while not self._stop:
            try:
                if self.lib.rsmi_dev_gpu_clk_freq_get(C.c_uint32(0), C.c_int(0), C.byref(f)) == 0 and f.current < 33:
                    self.mhz.append(int(f.frequency[f.current] // 1000000))
                if self.lib.rsmi_dev_current_socket_power_get(C.c_uint32(0), C.byref(pw)) == 0:
                    self.watts.append(pw.value / 1e6)
            except (AttributeError, OSError):
                break
            time.sleep(0.05)

    def __enter__(self):
        import threading
        if self.lib is not None:
            self._t = threading.Thread(target=self._run, daemon=True)
            self._t.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._t:
            self._t.join(timeout=1.0)

    def summary(self):
        if not self.mhz and not self.watts:
            return None
        r = {"samples": max(len(self.mhz), len(self.watts))}
        if self.mhz:
            r["sclk_mhz_median"] = int(np.median(self.mhz))
            r["sclk_mhz_min"] = int(min(self.mhz))
        if self.watts:
            r["socket_power_w_median"] = round(float(np.median(self.watts)), 0)
        return r

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
if (os.cpu_count() or 1) > 32:
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # cpu_baseline leg (OpenMP oracle) on a big shared host

# algorithmic work per unit: 2*M*N*K per Linear / conv, 4*Lq*Lk*H*d per attention
UNET_FWD_TFLOP = {"sd15": 0.803, "sdxl": 6.761, "sd35": 29.60, "flux": 69.47}
VAE_DECODE_TFLOP_1024 = 10.47  # KL-VAE decode 128x128 -> 1024x1024
MFMA_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="sd15", choices=["sd15", "sdxl", "sd15_tiny", "sd35", "sd35_tiny", "flux", "flux_tiny"])
    ap.add_argument("--batch", type=int, default=8, help="images per GPU (device batch)")
    ap.add_argument("--no-flash", action="store_true")
    ap.add_argument("--hip-graph", type=int, default=1, help="hipGraph replay of the step's plan (the product default); 0 = eager launches")
    ap.add_argument("--g16-variant", type=int, default=-1, help="gemm16 pipeline variant (A/B measurements)")
    ap.add_argument("--no-fuse-cfg", action="store_true", help="run cond and uncond as two graph computes (the reference's way)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--e2e", action="store_true", help="also time one full image (20 steps + VAE decode) per GPU batch (default on one GPU)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end sec/image leg")
    ap.add_argument("--backend-opt", action="append", default=[], metavar="KEY=INT",
                    help="planner / kernel option for A/B measurements, e.g. gemm16_sched=1 (ggml_backend_mi355x_set_option)")
    ap.add_argument("--device-sampler", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--host-loop", action="store_true",
                    help="time K host-driven cond+uncond forward pairs (x up / eps down per step, the reference's boundary) instead of the "
                         "device-resident trajectory")
    ap.add_argument("--no-sdxl", action="store_true", help="skip the SDXL 1024x1024 sub-record")
    ap.add_argument("--skip-legs", default="", help="comma list of sub-records to skip: sdxl,flux,sd35")
    ap.add_argument("--no-kernels", action="store_true", help="skip the per-kernel-family pass")
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="HARNESS SELF-CHECK, not a measurement: run the launch / sharding / reduction logic of this script on the CPU oracle backend "
                         "with gloo and the tiny UNet (tests/test_dist_shard.py::test_bench_gpus_2_launches_two_ranks)")
    return ap.parse_args()


def _file_version(path: Path) -> str:
    """Commit that last touched a committed profile (git checkout) or the sha1 of its bytes (the GPU box's snapshot has no .git)."""
    import hashlib
    import subprocess
    try:
        h = subprocess.run(["git", "log", "-1", "--format=%h", "--", str(path)], cwd=ROOT, capture_output=True, text=True, timeout=20).stdout.strip()
        if h:
            return "git:" + h
    except (OSError, subprocess.SubprocessError):
        pass
    return "sha1:" + hashlib.sha1(path.read_bytes()).hexdigest()[:12]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks (one process per GPU) under torch.distributed.run —
    the way the driver launches N > 1 itself.  Returns the launcher's exit code; rank 0's JSON line passes through on stdout."""
    import socket
    import subprocess

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    args.skip_legs = set(filter(None, args.skip_legs.split(",")))
    args.device_sampler = not args.host_loop
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or run `python bench.py --gpus N` and let it launch itself)")
    import torch
    import torch.distributed as dist

    import sdcpp_amd as sd

    selftest = args.selftest_cpu
    if selftest:
        # harness self-check on a CPU-only box: same launch / shard / reduce code, the oracle backend instead of the GPU, gloo instead of RCCL
        args.model, args.no_cpu_baseline, args.no_e2e, args.no_kernels, args.hip_graph = "sd15_tiny", True, True, True, 0
        args.batch = min(args.batch, 2)
        sd.load_backend(ROOT / "oracle" / "_build" / "libggml-cpu-oracle.so")
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product has no CPU fallback)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} has no GPU of its own (local_rank {local_rank}, {torch.cuda.device_count()} visible): one process per GPU")
        torch.cuda.set_device(local_rank)
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        sd.load_mi355x_backend()
    L = sd.lib()
    model_id = {"sd15": sd.SD15, "sdxl": sd.SDXL, "sd15_tiny": sd.SD15_TINY, "sd35": sd.SD35_LARGE, "sd35_tiny": sd.SD35_TINY,
                "flux": sd.FLUX_DEV, "flux_tiny": sd.FLUX_TINY}[args.model]
    backend_name = "CPU-oracle" if selftest else f"MI355X{local_rank}"   # one process per GPU: this rank's own device, never a shared one
    flux = args.model.startswith("flux")
    dit = args.model.startswith("sd35") or flux
    wtype = sd.Q8_0 if args.model == "sdxl" else (sd.Q4_0 if flux else (sd.BF16 if dit else sd.F16))   # BASELINE.json configs 3 / 4 / 5
    eng = sd.Engine(model=model_id, backend=backend_name, wtype=wtype, flash_attn=not args.no_flash)
    if not selftest:
        sd.backend_set_option("hip_graph", args.hip_graph)
        if args.g16_variant >= 0:
            sd.backend_set_option("gemm16_variant", args.g16_variant)
        for kv in args.backend_opt:
            k, v = kv.split("=")
            sd.backend_set_option(k.strip(), int(v))

    rng = np.random.default_rng(1234 + rank)
    tiny = args.model == "sd15_tiny"
    lat = 128 if args.model in ("sdxl", "sd35", "flux") else (16 if tiny or args.model in ("sd35_tiny", "flux_tiny") else 64)
    ctx_dim = {"sdxl": 2048, "sd35": 4096, "sd35_tiny": 96, "flux": 4096, "flux_tiny": 96}.get(args.model, 64 if tiny else 768)
    n_tok = 256 if flux else (154 if dit else 77)
    y_dim = {"sdxl": 2816, "sd35": 2048, "sd35_tiny": 64, "flux": 768, "flux_tiny": 64}.get(args.model)
    B = args.batch
    cond = rng.standard_normal((1, n_tok, ctx_dim)).astype(np.float32)
    uncond = np.random.default_rng(1235).standard_normal((1, n_tok, ctx_dim)).astype(np.float32)
    y = rng.standard_normal((1, y_dim)).astype(np.float32) if y_dim else None
    x = rng.standard_normal((B, 16 if dit else 4, lat, lat)).astype(np.float32)
    t = np.full((B,), 500.0, dtype=np.float32)

    fuse = not args.no_fuse_cfg and not flux   # FLUX.1-dev is guidance-distilled: cfg 1, ONE model call per step
    x2 = np.repeat(x, 2, axis=0)            # (b cond, b uncond, ...) interleaved
    t2 = np.repeat(t, 2)
    c2 = np.concatenate([cond, uncond], 0)  # [2,77,D]: tiled over the 2B images by the graph's ggml_repeat
    y2 = None if y is None else np.concatenate([y, y], 0)

    def step():
        # one sampler iteration's model work: cond + uncond forward over the device batch (host CFG/Euler math is included
        # in the e2e number; here the H2D/D2H crossings of the reference boundary are part of the step, as in the reference)
        if flux:
            eng.unet_forward(x, t * 0.0 + 0.5, cond, y)
        elif fuse:
            eng.unet_forward(x2, t2, c2, y2)
        else:
            eng.unet_forward(x, t, cond, y)
            eng.unet_forward(x, t, uncond, y)

    def dev_sync():
        if not selftest:
            torch.cuda.synchronize()

    def barrier():
        dev_sync()
        if world > 1:
            dist.barrier()
        dev_sync()

    def trajectory(k):
        # k sampler iterations on the device batch: x*c_in, cond+uncond forward, CFG, Euler(-A) update per step, no host crossing in between
        return eng.sample_latents(cond, None if flux else uncond, width=lat * 8, height=lat * 8, steps=k, cfg=1.0 if flux else 7.0, seed=42, batch=B,
                                  device_batch=B, method=sd.EULER if dit else sd.EULER_A, cond_y=y, uncond_y=y, fuse_cfg=True, device_sampler=True)

    if args.device_sampler:
        if args.warmup > 0:
            trajectory(args.warmup)
    else:
        for _ in range(args.warmup):
            step()
    barrier()
    # Eager launches (hip_graph 0): the dominant family's HIP events are recorded INSIDE the timed region.  hipGraph replay (the product default): events
    # recorded inside a captured graph cannot be read back on this runtime (scripts/graph_event_probe.hip), so the timed region runs untouched and the
    # same K steps are repeated eagerly right after it with the events on (roofline.timing says which; eager_ms_per_step is that pass's wall time).
    timing = not selftest
    timing_live = timing and args.hip_graph == 0
    if timing_live:
        sd.kernel_timing_enable(True)
    sclk = SclkSampler()
    t0 = time.perf_counter()
    with sclk:
        if args.device_sampler:
            timed_latents = trajectory(args.steps)
        else:
            for _ in range(args.steps):
                step()
        barrier()
    dt = time.perf_counter() - t0
    if args.device_sampler and not np.isfinite(timed_latents).all():   # a timing on NaN data is not a measurement
        raise RuntimeError(f"bench: the latents sampled in the timed region are not finite ({int((~np.isfinite(timed_latents)).sum())} of {timed_latents.size} values)")
    kt = sd.kernel_timing() if timing_live else None
    if timing_live:
        sd.kernel_timing_enable(False)
    eager_ms = None
    if timing and not timing_live:
        sd.kernel_timing_enable(True)   # plans run eagerly while a family is being timed
        barrier()
        t1 = time.perf_counter()
        if args.device_sampler:
            trajectory(args.steps)
        else:
            for _ in range(args.steps):
                step()
        barrier()
        eager_ms = (time.perf_counter() - t1) / args.steps * 1e3
        kt = sd.kernel_timing()
        sd.kernel_timing_enable(False)
    per_rank_ms = [round(dt / args.steps * 1e3, 3)]
    rccl_ranks = 1
    if world > 1:
        tt = torch.zeros(world + 1, device="cpu" if selftest else "cuda", dtype=torch.float64)
        tt[rank] = dt
        tt[world] = 1.0   # one per rank that reached the reduction: n_gpus is what actually ran, not what was asked for
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(v) / args.steps * 1e3, 3) for v in tt[:world].tolist()]   # stragglers show up here
        dt = float(tt[:world].max().item())
        rccl_ranks = int(round(float(tt[world].item())))
    ms_per_step = dt / args.steps * 1e3
    its = B * world * args.steps / dt

    fwd_tflop = UNET_FWD_TFLOP.get(args.model, 0.0)
    step_tflops = ((1 if flux else 2) * B * fwd_tflop) / (ms_per_step / 1e3) if fwd_tflop else 0.0
    roofline = {"bound": "mfma", "achieved": None, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None}
    if kt and kt["launches"] > 0 and kt["total_ms"] > 0:
        achieved = kt["total_flops"] / (kt["total_ms"] * 1e-3) / 1e12
        roofline.update({"achieved": round(achieved, 2), "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "kernel": kt["kernel"],
                         "launches": kt["launches"], "avg_launch_us": round(kt["total_ms"] * 1e3 / kt["launches"], 2),
                         "avg_launch_gflop": round(kt["total_flops"] / kt["launches"] / 1e9, 2),
                         # algorithmic HBM bytes of the same launches: NHWC f16 input image + weight image + f32 output (+ residual), each once
                         "avg_launch_algorithmic_bytes": round(kt["total_bytes"] / kt["launches"]),
                         "share_of_step_time": round(kt["total_ms"] / (dt * 1e3), 4),
                         "timing": ("hipEventElapsedTime around each launch on the launch stream, inside the timed region" if timing_live else
                                    "hipEventElapsedTime around each launch on the launch stream, over the same K steps repeated EAGERLY right after the timed "
                                    "region (the timed region replays a hipGraph; events inside a captured graph cannot be read back)")})
        if eager_ms is not None:
            roofline["eager_ms_per_step"] = round(eager_ms, 3)
            roofline["share_of_step_time"] = round(kt["total_ms"] / (eager_ms * args.steps), 4)
        # PMC passes over THIS kernel family (scripts/gpu_round_end3.sh); the newest committed set: profiles/r<round><call>_pmc_traffic_conv256.json
        tfs = sorted((ROOT / "profiles").glob("r[0-9][0-9][a-zA-Z]_pmc_traffic_conv256.json"))
        tf = tfs[-1] if tfs else ROOT / "profiles" / "r04_pmc_traffic_conv256.json"
        if tf.exists() and args.model == "sd15" and B == 8 and fuse:
            try:
                pm = json.loads(tf.read_text())
                roofline["traffic"] = pm["hbm_bytes_per_launch"]
                roofline["traffic_measured_in_this_run"] = False  # rocprofv3 --pmc cannot run inside this process: a committed profile of the same command
                roofline["traffic_commit"] = _file_version(tf)
                roofline["traffic_source"] = "profiles/" + tf.name + " (kernels matching '" + pm.get("kernel", "") + "', " + str(pm.get("launches_fetch_pass")) + " launches): " + pm.get("source", "")
                if pm.get("note"):
                    roofline["traffic_note"] = pm["note"]
            except (ValueError, KeyError):
                pass
    if not selftest and rank == 0:
        # what THIS box delivers (csrc/kernels/calib.hip, ~1 s, after the timed region): the pool's boxes differ by +-7 % with one binary, so a line is only
        # comparable with another line through these
        cal = sd.calibrate()
        if cal:
            roofline["measured_peaks"] = {**cal, "note": "MFMA loop from registers (v_mfma_f32_32x32x16_f16, 8 waves per CU), float4 copy / read of 1 GiB, shader clock during "
                                                         "the MFMA loop; measured on this box right after the timed region"}
            if roofline.get("achieved"):
                roofline["frac_of_measured_mfma"] = round(roofline["achieved"] / cal["mfma_f16_tflops"], 4)
            roofline["value_per_measured_mfma_pflops"] = round(its / (cal["mfma_f16_tflops"] / 1e3), 2)   # it/s per measured PFLOP/s: the box-independent form of `value`
        ss = sclk.summary()
        if ss:   # the clock the chip actually held during the timed steps (power-capped: the nominal MFMA peak assumes 2.4 GHz)
            ss["mfma_peak_at_that_clock_tflops"] = round(MFMA_PEAK_TFLOPS * ss["sclk_mhz_median"] / 2400.0, 1) if "sclk_mhz_median" in ss else None
            if ss.get("mfma_peak_at_that_clock_tflops") and roofline.get("achieved"):
                ss["dominant_family_frac_of_that"] = round(roofline["achieved"] / ss["mfma_peak_at_that_clock_tflops"], 4)
            roofline["sustained_clock"] = ss
    roofline["whole_step_tflops"] = round(step_tflops, 2)  # all kernels + host graph build + uploads: 2*B*UNet-forward FLOPs / step wall time
    roofline["whole_step_frac"] = round(step_tflops / MFMA_PEAK_TFLOPS, 4)
    if timing and rank == 0 and not args.no_kernels:
        roofline["kernels"] = kernel_families(sd, trajectory if args.device_sampler else None, step)
    out = {
        "metric": ("SELFTEST on the CPU oracle (harness check, NOT a measurement) — " if selftest else "") + "denoise it/s (image-iterations/s: cond+uncond UNet forwards per image per step)",
        "value": round(its, 3),
        "unit": "it/s",
        "n_gpus": rccl_ranks,
        "rccl_ranks": rccl_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "ms_per_step_per_rank": per_rank_ms,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f16",
        "data": "synthetic",
        "devices": [backend_name] if world == 1 else f"one process per GPU: rank r drives MI355X<r> ({world} ranks)",
        "config": {"workload": f"{args.model} {'MMDiT' if dit else 'UNet'} {lat*8}x{lat*8}, {'cfg 1 (distilled guidance 3.5, one forward per step)' if flux else 'cfg 7 (cond+uncond)'}, {'q8_0 Linear + f16 conv' if args.model == 'sdxl' else ('q4_0' if flux else ('bf16' if dit else 'f16'))} weights, batch {B}/GPU, Euler-A step",
                   "global_batch": B * world, "flash_attn": not args.no_flash, "hip_graph": args.hip_graph,
                   "cfg_pair_in_one_graph": fuse, "device_resident_sampler": bool(args.device_sampler),
                   **({"backend_opts": args.backend_opt} if args.backend_opt else {})},
        "roofline": roofline,
    }
    # BASELINE.json's metric is "denoise it/s + sec/image": the second half is one whole sdm_generate_image call (noise -> 20 sampler steps ->
    # VAE decode -> uint8 pixels) on the same device batch, run AFTER the timed region.  Default on one GPU; --e2e forces it on rank 0.
    if rank == 0 and not args.no_e2e and (args.e2e or world == 1):
        try:
            eng.vae_decode(np.zeros((B, 16 if dit else 4, lat, lat), dtype=np.float32))  # untimed: builds the VAE weight images and plan
            t0 = time.perf_counter()
            eng.generate_image(cond, None if flux else uncond, width=lat * 8, height=lat * 8, steps=20, cfg=1.0 if flux else 7.0, seed=42, batch=B,
                               device_batch=B, method=sd.EULER if dit else sd.EULER_A, cond_y=y, uncond_y=y, fuse_cfg=fuse,
                               device_sampler=args.device_sampler)
            e2e = time.perf_counter() - t0
            st = eng.stats()
            out["e2e"] = {"sec_per_image": round(e2e / B, 4), "batch": B, "steps": 20, "sample_ms": round(st["last_sample_ms"], 1),
                          "vae_decode_ms": round(st["last_decode_ms"], 1)}
        except Exception as exc:  # the headline line must survive a failure of the extra leg
            out["e2e"] = {"error": str(exc)[:200]}
    if rank == 0 and world == 1 and args.device_sampler and not selftest:
        # the PCIe-inclusive rate of the reference's boundary (host buffers in and out every step), for DESIGN.md — never the headline
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        hl = (time.perf_counter() - t0) / 5
        out["host_loop"] = {"ms_per_step": round(hl * 1e3, 3), "it_per_s": round(B / hl, 2),
                            "note": "cond+uncond forward pair per step from host buffers (x H2D, eps D2H each step); CFG / Euler math on the host is outside this step"}
    if rank == 0 and world == 1 and args.model == "sd15":
        for leg in ("sdxl", "flux", "sd35", "sdxl_b8"):
            if leg in args.skip_legs or (leg == "sdxl" and args.no_sdxl):
                continue
            try:
                out[leg] = model_leg(sd, backend_name, args, leg)
            except Exception as exc:
                out[leg] = {"error": str(exc)[:200]}
    if world > 1 and args.model in ("sd15", "sd15_tiny") and "multi" not in args.skip_legs:
        # the metric names SDXL 1024x1024 at 2 / 4 / 8 GPUs too, and the one real exchange step of the path: every rank takes part (collectives inside)
        multi = multi_gpu_legs(sd, backend_name, args, dist, rank, world, selftest)
        if rank == 0:
            out.update(multi)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dit:  # the CPU leg is defined for the headline UNet workloads
        out["cpu_baseline"] = cpu_baseline(sd, args, lat, ctx_dim)
    if rank == 0 and selftest:
        print(json.dumps(out), flush=True)
    elif rank == 0:
        st = sd.backend_stats()
        out["backend"] = {k: st[k] for k in ("swizzled_weight_bytes", "qgemv_linears", "fgemv_linears", "fused_presilu", "fused_sibling_linears", "hoisted_kv_linears",
                                             "qgemm16_linears", "jit_images", "qinloop_linears", "flash_out_alias", "flash_slice_images", "split_k_gemms", "fused_attention", "generic_matmul", "plans_built", "graph_replays")}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def kernel_families(sd, trajectory, step):
    """Every kernel family of one step, timed with HIP events around each dispatch in a SEPARATE short pass (2 steps)."""
    if trajectory is not None:
        trajectory(1)
    else:
        step()
    sd.kernel_timing_enable(sd.KF_ALL)
    n = 2
    if trajectory is not None:
        trajectory(n)
    else:
        for _ in range(n):
            step()
    fams = sd.kernel_timings()
    sd.kernel_timing_enable(0)
    tot = sum(f["total_ms"] for f in fams) or 1.0
    rows = []
    for f in sorted(fams, key=lambda f: -f["total_ms"]):
        sec = f["total_ms"] * 1e-3
        if f["bound"] == "mfma":
            ach, peak, unit = f["total_flops"] / sec / 1e12, MFMA_PEAK_TFLOPS, "TFLOP/s"
        else:
            ach, peak, unit = f["total_bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
        rows.append({"name": f["kernel"], "bound": f["bound"], "launches_per_step": round(f["launches"] / n, 1), "ms_per_step": round(f["total_ms"] / n, 3),
                     "share_of_step_time": round(f["total_ms"] / tot, 4), "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4)})
        if f["bound"] == "mfma" and f["total_bytes"] > 0:  # contraction families: also the algorithmic-byte rate (operands once, output once) vs HBM
            rows[-1]["algorithmic_gbs"] = round(f["total_bytes"] / sec / 1e9, 1)
    return rows


def multi_gpu_legs(sd, backend_name, args, dist, rank, world, selftest):
    """N > 1 only, after the headline.  (1) `sdxl_sharded`: BASELINE.json config 3 as it shards — ONE SDXL 1024x1024 image per rank (q8_0, cfg 7, Euler-A,
    device-resident), no data-path collective; value = images x steps / max-over-ranks time.  (2) `sdxl_cfg_pair_split`: ranks 0 and 1 run ONE image as a CFG
    pair — cond on rank 0, uncond on rank 1, one 2-rank all-reduce of the pre-scaled eps per step (RCCL over one xGMI link; src/runtime/guidance.cpp:149-179 on two
    devices) — with the all-reduce of an eps-sized buffer timed on its own.  The harness self-check runs the same code on the tiny UNet
    over gloo."""
    import torch

    from sdcpp_amd import shard

    k = 4 if selftest else 8
    if selftest:
        model, wtype, lat, cdim, ydim = sd.SD15_TINY, sd.F16, 16, 64, None
    else:
        model, wtype, lat, cdim, ydim = sd.SDXL, sd.Q8_0, 128, 2048, 2816
    eng = sd.Engine(model=model, backend=backend_name, wtype=wtype, flash_attn=not args.no_flash)
    rng = np.random.default_rng(99)
    cond = rng.standard_normal((1, 77, cdim)).astype(np.float32)
    uncond = rng.standard_normal((1, 77, cdim)).astype(np.float32)
    y = rng.standard_normal((1, ydim)).astype(np.float32) if ydim else None
    kw = dict(width=lat * 8, height=lat * 8, cfg=7.0, cond_y=y, uncond_y=y)
    dev = "cpu" if selftest else "cuda"

    def sync():
        if not selftest:
            torch.cuda.synchronize()
        dist.barrier()
        if not selftest:
            torch.cuda.synchronize()

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    run = lambda steps: eng.sample_latents(cond, uncond, steps=steps, seed=42 + rank, batch=1, device_batch=1, method=sd.EULER_A, fuse_cfg=True, device_sampler=True, **kw)
    run(1)
    sync()
    t0 = time.perf_counter()
    run(k)
    sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    res = {"sdxl_sharded": {"workload": ("SELFTEST tiny UNet" if selftest else "sdxl 1024x1024, q8_0, cfg 7 (cond+uncond in one graph), Euler-A, device-resident") +
                                        f": 1 image per rank x {world} ranks (BASELINE.json config 3 sharded), no data-path collective",
                            "ranks": world, "steps_timed": k, "ms_per_step": round(dt / k * 1e3, 3), "it_per_s": round(world * k / dt, 3), "scaling": "weak"}}
    # ---- CFG-pair split on ranks 0 / 1 (every rank creates the group: new_group is collective)
    pair = dist.new_group([0, 1])
    if rank < 2:
        split = lambda steps: shard.sample_cfg_pair_split(eng, cond, uncond, steps=steps, seed=42, dist=dist, group=pair, rank_in_pair=rank, batch=1, **kw)
        split(1)
    sync()
    t0 = time.perf_counter()
    if rank < 2:
        lat_split = split(k)
    sync()
    dt_split = max_over_ranks(time.perf_counter() - t0)
    ex_us = None
    if rank < 2:
        n_eps = (4 * lat * lat)
        buf = torch.zeros(n_eps, dtype=torch.float32, device=dev)
        for _ in range(3):
            dist.all_reduce(buf, group=pair)
        if not selftest:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            dist.all_reduce(buf, group=pair)
        if not selftest:
            torch.cuda.synchronize()
        ex_us = (time.perf_counter() - t0) / 20 * 1e6
    same = None
    if rank < 2:   # both ranks of the pair must hold the same latents (same update, same Philox noise, summed eps)
        chk = torch.from_numpy(np.ascontiguousarray(lat_split, dtype=np.float32)).to(dev)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=pair)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=pair)
        same = bool(torch.equal(lo, hi))
    res["sdxl_cfg_pair_split"] = {"workload": "ONE image on ranks 0 + 1: cond forward on rank 0, uncond on rank 1, one 2-rank all-reduce (SUM) of the pre-scaled eps per step "
                                              f"({'gloo' if selftest else 'RCCL'}), same Euler-A update on both",
                                  "steps_timed": k, "ms_per_step": round(dt_split / k * 1e3, 3), "eps_floats": 4 * lat * lat,
                                  "exchange_us_per_step": None if ex_us is None else round(ex_us, 1), "latents_identical_on_both_ranks": same,
                                  "vs_one_gpu_pair_ms_per_step": res["sdxl_sharded"]["ms_per_step"]}
    del eng
    return res


LEGS = {
    # name: (model attr, weight type attr, latent, ctx tokens, ctx dim, y dim, latent channels, per-GPU batch, timed steps, sampler steps of the config, cfg, forwards per step)
    "sdxl": ("SDXL", "Q8_0", 128, 77, 2048, 2816, 4, 1, 8, 30, 7.0, 2),       # config 3: batch 8 over 8 GPUs -> 1 image per GPU
    "flux": ("FLUX_DEV", "Q4_0", 128, 256, 4096, 768, 16, 1, 5, 28, 1.0, 1),   # config 4: one GPU, cfg 1 (distilled guidance 3.5)
    "sd35": ("SD35_LARGE", "BF16", 128, 154, 4096, 2048, 16, 2, 5, 28, 7.0, 2),  # config 5: batch 16 over 8 GPUs -> 2 images per GPU
    "sdxl_b8": ("SDXL", "Q8_0", 128, 77, 2048, 2816, 4, 8, 5, 30, 7.0, 2),    # config 3 at its ONE-GPU point: the whole batch of 8 on one MI355X
}


def model_leg(sd, backend_name, args, name):
    """The other workloads BASELINE.json names, on this GPU's share of the configuration, a few device-resident sampler steps after the timed
    region (bounded: the default bench run must finish in minutes):
      sdxl  SDXL UNet 1024x1024, q8_0 Linear + f16 conv, cfg 7 (cond+uncond in one graph), batch 1, Euler-A; plus the 1024x1024 VAE decode, so
            that sec_per_image is the true end-to-end figure (30 steps + decode);
      flux  FLUX.1-dev 1024x1024 (4096 + 256 tokens), q4_0, cfg 1, batch 1, Euler;
      sd35  SD3.5-large 1024x1024 (4096 + 154 tokens), bf16, cfg 7, batch 2, Euler.
    Each carries ms_per_step, it/s, the whole-step fraction of the MFMA peak and its three heaviest kernel families (HIP events)."""
    import torch

    mattr, wattr, lat, ntok, cdim, ydim, ch, B, k, cfg_steps, cfg, nfwd = LEGS[name]
    dit = not name.startswith("sdxl")
    t_init = time.perf_counter()
    eng = sd.Engine(model=getattr(sd, mattr), backend=backend_name, wtype=getattr(sd, wattr), flash_attn=not args.no_flash)
    init_s = time.perf_counter() - t_init
    rng = np.random.default_rng(99)
    cond = rng.standard_normal((1, ntok, cdim)).astype(np.float32)
    uncond = rng.standard_normal((1, ntok, cdim)).astype(np.float32)
    y = rng.standard_normal((1, ydim)).astype(np.float32)
    kw = dict(width=lat * 8, height=lat * 8, cfg=cfg, seed=42, batch=B, device_batch=B, method=sd.EULER if dit else sd.EULER_A, cond_y=y, uncond_y=y,
              fuse_cfg=True, device_sampler=True)
    unc = None if nfwd == 1 else uncond
    st0 = sd.backend_stats()
    eng.sample_latents(cond, unc, steps=1, **kw)   # builds weight images + plan
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lat_out = eng.sample_latents(cond, unc, steps=k, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / k * 1e3
    # a timing on NaN data is not a measurement (round 6 found the batch >= 2 DiT legs of earlier rounds running on NaNs: an operand-aliasing race in a fusion)
    if not np.isfinite(lat_out).all():
        raise RuntimeError(f"bench leg {name}: the sampled latents are not finite ({int((~np.isfinite(lat_out)).sum())} of {lat_out.size} values)")
    tfl = nfwd * B * UNET_FWD_TFLOP[name.split("_")[0]] / (ms / 1e3)
    res = {"workload": f"{name.split('_')[0]} 1024x1024, {'cfg 1 (one forward per step)' if nfwd == 1 else 'cfg 7 (cond+uncond in one graph)'}, {wattr.lower()} Linear weights, "
                       f"batch {B}/GPU, {'Euler' if dit else 'Euler-A'} step, device-resident",
           "steps_timed": k, "ms_per_step": round(ms, 2), "it_per_s": round(B * 1e3 / ms, 3), "whole_step_tflops": round(tfl, 1),
           "whole_step_frac": round(tfl / MFMA_PEAK_TFLOPS, 4), "engine_init_s": round(init_s, 1)}
    if not args.no_kernels:
        sd.kernel_timing_enable(sd.KF_ALL)
        eng.sample_latents(cond, unc, steps=1, **kw)
        fams = sd.kernel_timings()
        sd.kernel_timing_enable(0)
        tot = sum(f["total_ms"] for f in fams) or 1.0
        top = []
        for f in sorted(fams, key=lambda f: -f["total_ms"])[:3]:
            sec = f["total_ms"] * 1e-3
            mf = f["bound"] == "mfma"
            ach = (f["total_flops"] / sec / 1e12) if mf else (f["total_bytes"] / sec / 1e9)
            top.append({"name": f["kernel"], "bound": f["bound"], "launches_per_step": f["launches"], "ms_per_step": round(f["total_ms"], 3),
                        "share_of_step_time": round(f["total_ms"] / tot, 4), "achieved": round(ach, 1), "unit": "TFLOP/s" if mf else "GB/s",
                        "frac": round(ach / (MFMA_PEAK_TFLOPS if mf else HBM_PEAK_GBS), 4)})
        res["kernels"] = top
        res["kernel_ms_per_step"] = round(tot, 2)
    # the end-to-end half of the metric, TIMED for every leg (VERDICT r4 missing #3, r5 missing #4): the VAE decode on its own, then one sdm_generate_image call
    # = noise -> the configuration's sampler steps -> 128x128 -> 1024x1024 KL-VAE decode (10.47 TFLOP per image; 4-channel with Conv2d scale 1/32 for SDXL like
    # the reference without --vae, 16-channel for SD3.5 / FLUX: config 5's "full VAE decode (no TAESD)") -> u8 pixels on the host
    z = rng.standard_normal((B, ch, lat, lat)).astype(np.float32)
    eng.vae_decode(z)   # untimed: builds the VAE weight images and plan
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.vae_decode(z)
    torch.cuda.synchronize()
    dec = (time.perf_counter() - t0) * 1e3
    res["vae_decode_ms"] = round(dec, 1)
    res["vae_decode_frac"] = round(VAE_DECODE_TFLOP_1024 * B / (dec / 1e3) / MFMA_PEAK_TFLOPS, 4)
    try:  # the tiny autoencoder's decode of the same latents (the reference's --taesd; reported beside the KL-VAE decode, not part of sec_per_image)
        eng.tae_decode(z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.tae_decode(z)
        torch.cuda.synchronize()
        res["taesd_decode_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
    except Exception as ex:  # never let the side measurement take the leg down
        res["taesd_decode_ms"] = None
        res["taesd_error"] = str(ex)[:200]
    t0 = time.perf_counter()
    img = eng.generate_image(cond, unc, steps=cfg_steps, **kw)
    e2e = time.perf_counter() - t0
    st = eng.stats()
    assert img.shape == (B, lat * 8, lat * 8, 3)
    res["sec_per_image"] = round(e2e / B, 4)
    res["e2e"] = {"timed": True, "batch": B, "steps": cfg_steps, "wall_s": round(e2e, 3), "sample_ms": round(st["last_sample_ms"], 1),
                  "vae_decode_ms": round(st["last_decode_ms"], 1), "note": "one sdm_generate_image call: noise, sampler steps, VAE decode, u8 conversion"}
    st1 = sd.backend_stats()
    res["weight_image_bytes"] = st1["swizzled_weight_bytes"] - st0["swizzled_weight_bytes"]   # f16 images kept resident (cached) for this model
    res["jit_image_linears"] = st1["jit_images"] - st0["jit_images"]   # quantised Linears planned WITHOUT a resident image (rebuilt per launch, option jit_qimages)
    res["inloop_dequant_linears"] = st1["qinloop_linears"] - st0["qinloop_linears"]   # ... and those whose GEMM reads the raw GGUF blocks and dequantises them in its main loop (no image at all)
    res["flash_out_alias"] = st1["flash_out_alias"] - st0["flash_out_alias"]   # attention outputs NOT written in their final layout by the flash kernel because the allocator gave that buffer an operand's block
    res["raw_block_few_row_linears"] = (st1["qgemm16_linears"] - st0["qgemm16_linears"]) + (st1["qgemv_linears"] - st0["qgemv_linears"])   # k_qgemm16 / k_qgemv plans (<= 512 rows)
    del eng
    return res


def host_cpu_info():
    """CPU model and core counts of the box ("core count stated"): /proc/cpuinfo, no subprocess needed."""
    info = {"logical_cpus": os.cpu_count()}
    try:
        model, phys = None, set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                pid = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":", 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
        info["model"] = model
        info["physical_cores"] = len(phys) or None
        info["schedulable_cpus"] = len(os.sched_getaffinity(0))
        try:   # the container's CPU-time quota: what actually bounds an OpenMP team here (the GPU boxes of this pool: 16 CPUs of a 256-thread host)
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            info["cgroup_cpu_quota"] = None if quota == "max" else round(int(quota) / int(period), 2)
        except (OSError, ValueError):
            info["cgroup_cpu_quota"] = None
    except OSError:
        pass
    return info


def cpu_baseline(sd, args, lat, ctx_dim):
    """Time the CPU oracle (checker, never the product path) on a bounded sample: UNet forwards of ONE image at the same
    resolution; converted to the metric's unit (one it = cond + uncond forward)."""
    oracle_so = ROOT / "oracle" / "_build" / "libggml-cpu-oracle.so"
    if not oracle_so.exists():
        return {"value": None, "unit": "it/s", "cores": 0, "kind": "port", "sample": "oracle library not built"}
    sd.load_backend(oracle_so)
    olib = C.CDLL(str(oracle_so))
    cores = int(olib.oracle_num_threads())
    host = host_cpu_info()
    model_id = {"sd15": sd.SD15, "sdxl": sd.SDXL, "sd15_tiny": sd.SD15_TINY, "sd35": sd.SD35_LARGE, "sd35_tiny": sd.SD35_TINY}[args.model]
    eng = sd.Engine(model=model_id, backend="CPU-oracle", wtype=sd.Q8_0 if args.model == "sdxl" else sd.F16, flash_attn=False)
    rng = np.random.default_rng(7)
    x = rng.standard_normal((1, 4, lat, lat)).astype(np.float32)
    t = np.full((1,), 500.0, dtype=np.float32)
    ctx = rng.standard_normal((1, 77, ctx_dim)).astype(np.float32)
    y = rng.standard_normal((1, 2816)).astype(np.float32) if args.model == "sdxl" else None
    olib.oracle_set_num_threads.restype = C.c_int

    def timed(budget_s):
        n = 0
        t0 = time.perf_counter()
        while True:
            eng.unet_forward(x, t, ctx, y)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s or n >= 8:
                return n, el

    # two team sizes: the oracle's default (<= 16 threads: the team the parity tests use) and every CPU this process may be
    # scheduled on (affinity mask and cgroup quota).  `value` / `cores` = the faster of the two; both are listed.
    runs = []
    n, el = timed(args.cpu_baseline_seconds / 2)
    runs.append({"cores": cores, "it_per_s": round(n / (2 * el), 5), "sec_per_forward": round(el / n, 3), "forwards": n})
    all_cpus = int(olib.oracle_set_num_threads(0))   # affinity mask AND the cgroup CPU quota: threads beyond the quota only time-slice
    if all_cpus > cores:
        eng.unet_forward(x, t, ctx, y)  # untimed: the bigger team's threads start
        n2, el2 = timed(args.cpu_baseline_seconds / 2)
        runs.append({"cores": all_cpus, "it_per_s": round(n2 / (2 * el2), 5), "sec_per_forward": round(el2 / n2, 3), "forwards": n2})
    olib.oracle_set_num_threads(cores)
    best = max(runs, key=lambda r: r["it_per_s"])
    return {"value": best["it_per_s"], "unit": "it/s", "cores": best["cores"], "kind": "port", "host": host, "team_sizes": runs,
            "sample": f"{best['forwards']} UNet forward(s) of 1 image ({args.model}, latent {lat}x{lat}) on {best['cores']} threads; one it = 2 forwards (cfg 7); "
                      f"`cores` = the OpenMP team = every CPU this container may use ({all_cpus}: affinity mask and cgroup quota "
                      f"{host.get('cgroup_cpu_quota')}) of a host with {host.get('physical_cores')} physical cores",
            "sec_per_forward": best["sec_per_forward"]}


if __name__ == "__main__":
    main()
