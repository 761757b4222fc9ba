#!/bin/bash
# round 4, GPU call 2: parity suite with the new kernels (prefill attention, fused first Downsampler conv, x-reuse convs),
# per-kernel benches, short bench line, kernel trace with the in-situ numbers of the PROFILED run
cd "$(dirname "$0")/.."
O=gpurun_out/r4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_call2.txt 2>&1; echo "pytest rc $?" >> $O/pytest_call2.txt
tail -n 8 $O/pytest_call2.txt
{ timeout 300 python tools/kbench_conv.py --xreuse 1; timeout 300 python tools/kbench_conv.py --xreuse 0; } > $O/kbench_conv_call2.txt 2>&1
tail -n 40 $O/kbench_conv_call2.txt
timeout 300 python tools/kbench_enc.py > $O/kbench_enc_call2.txt 2>&1; cat $O/kbench_enc_call2.txt
timeout 900 python bench.py --steps 4 --warmup 1 --no-subrecords --no-cpu-baseline > $O/bench_call2.json 2> $O/bench_call2.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench_call2.json"))
print({k: d[k] for k in ("value", "ms_per_step", "stages_ms", "sanity")})
print(json.dumps(d["roofline"])[:1500])
for k in d["kernels"]: print(k["kernel"][:70], k["ms"], k["frac"])
PY
PROF_OUT=$PWD/$O timeout 900 tools/prof_run.sh bench_trace2 python $PWD/bench.py --steps 2 --warmup 1 --no-subrecords --no-cpu-baseline --no-kernels > /dev/null 2>&1
head -n 16 $O/prof_bench_trace2.txt | cut -c1-1200
