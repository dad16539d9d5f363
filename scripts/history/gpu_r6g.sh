# round 5: full per-family kernel tables (roofline.kernels[]) of the other configurations, each as its own bench.py headline run on the final code
D=gpurun_out/r6g
mkdir -p $D
export OMP_WAIT_POLICY=PASSIVE
timeout 500 python bench.py --model sdxl --batch 8 --steps 4 --warmup 1 --no-cpu-baseline --no-e2e > $D/bench_sdxl_b8.jsonl 2> $D/err_sdxl_b8.log; echo "sdxl b8 rc=$?"
timeout 500 python bench.py --model sdxl --batch 1 --steps 8 --warmup 2 --no-cpu-baseline --no-e2e > $D/bench_sdxl_b1.jsonl 2> $D/err_sdxl_b1.log; echo "sdxl b1 rc=$?"
timeout 500 python bench.py --model flux --batch 1 --steps 4 --warmup 1 --no-cpu-baseline --no-e2e > $D/bench_flux.jsonl 2> $D/err_flux.log; echo "flux rc=$?"
timeout 500 python bench.py --model sd35 --batch 2 --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > $D/bench_sd35.jsonl 2> $D/err_sd35.log; echo "sd35 rc=$?"
for f in sdxl_b8 sdxl_b1 flux sd35; do python - "$D/bench_$f.jsonl" <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith("{"):
        p = json.loads(l); r = p["roofline"]
        print(sys.argv[1].split("/")[-1], p["value"], "it/s", p["ms_per_step"], "ms/step; whole step", r["whole_step_frac"])
        for k in r.get("kernels", [])[:6]:
            print(f"   {k['name'][:60]:60s} {k['launches_per_step']:7.1f} {k['ms_per_step']:8.3f} ms {k['share_of_step_time']*100:5.1f}% {k['achieved']:8.1f} {k['unit']} {k['frac']*100:5.1f}%")
PY
done
