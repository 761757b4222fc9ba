#!/bin/bash
# round 4, GPU call 3: parity suite (DPP attention reductions, encoder interior runs, batched sparse Downsampler conv),
# encoder bench, AR-loop sweep, PMC traffic of the decode kernels at 96 rows, short bench line
cd "$(dirname "$0")/.."
O=gpurun_out/r4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_call3.txt 2>&1; echo "pytest rc $?" >> $O/pytest_call3.txt
tail -n 8 $O/pytest_call3.txt
timeout 300 python tools/kbench_enc.py > $O/kbench_enc_call3.txt 2>&1; cat $O/kbench_enc_call3.txt
timeout 600 python tools/ar_sweep.py --out $O/ar_sweep_call3.txt --reps 2 < tools/sweeps/r04_call3.txt > $O/ar_sweep_call3.log 2>&1; cat $O/ar_sweep_call3.txt | cut -c1-260
export TMPDIR=/tmp
timeout 600 python tools/pmc_traffic.py 96 $O/r04_pmc_traffic_B96.json 2>&1 | tail -n 3
timeout 900 python bench.py --steps 3 --warmup 1 --no-subrecords --no-cpu-baseline --no-kernels > $O/bench_call3.json 2> $O/bench_call3.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench_call3.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms", "turnstile")})
PY
