#!/bin/bash
# round 4, GPU call 5: parity suite (prefill attention on operand-order LDS rows, RCCL rs_ag leg), the perf-marked test, kernel lines
cd "$(dirname "$0")/.."
O=gpurun_out/r4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_call5.txt 2>&1; echo "pytest rc $?" >> $O/pytest_call5.txt
tail -n 8 $O/pytest_call5.txt
timeout 300 python -m pytest tests/test_perf_gpu.py -m perf -q > $O/pytest_perf_call5.txt 2>&1; tail -n 3 $O/pytest_perf_call5.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-subrecords --no-cpu-baseline > $O/bench_call5.json 2> $O/bench_call5.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench_call5.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms", "turnstile")})
print(d["sanity"])
for k in d["kernels"]: print(k["kernel"][:70], k["ms"], k["frac"])
PY
