#!/bin/bash
# round 4, final records: parity suite, the driver's own bench invocation, the kernel trace of a short run of the same command,
# the kernel trace of encode_cl
cd "$(dirname "$0")/.."
O=gpurun_out/r4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_final.txt 2>&1; echo "pytest rc $?" >> $O/pytest_final.txt; tail -n 4 $O/pytest_final.txt
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_like.json 2> $O/bench_driver_like.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench_driver_like.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms", "turnstile", "sanity")})
print(d["roofline"]["summary"])
for k in d["kernels"]: print(k["kernel"][:70], k["ms"], k["frac"])
print(d["config3"]["batch16_whole_path"], d["config2"]["shapes_per_s"], d["config4"]["shapes_per_s"])
PY
PROF_OUT=$PWD/$O timeout 900 tools/prof_run.sh bench_trace_final python $PWD/bench.py --steps 2 --warmup 1 --no-subrecords --no-cpu-baseline --no-kernels > /dev/null 2>&1
head -n 12 $O/prof_bench_trace_final.txt | cut -c1-400
PROF_OUT=$PWD/$O tools/prof_run.sh enc_real python $PWD/tools/prof_enc_only.py real > /dev/null 2>&1; head -n 12 $O/prof_enc_real.txt | cut -c1-120
