#!/bin/bash
# round 4, final records: the driver's own bench invocation, and the kernel trace of a short run of the same command
cd "$(dirname "$0")/.."
O=gpurun_out/r4; mkdir -p $O
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench_driver_like.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms", "turnstile", "sanity")})
print(json.dumps(d["roofline"])[:2500])
print(json.dumps(d["config3"])[:900])
print(json.dumps(d.get("config2"))[:300]); print(json.dumps(d.get("config4"))[:300]); print(json.dumps(d.get("train"))[:500])
print(d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_kv", {}).get("value"))
PY
PROF_OUT=$PWD/$O timeout 900 tools/prof_run.sh bench_trace_final python $PWD/bench.py --steps 2 --warmup 1 --no-subrecords --no-cpu-baseline --no-kernels > /dev/null 2>&1
head -n 14 $O/prof_bench_trace_final.txt | cut -c1-700
