#!/bin/bash
# round 4, GPU call 1: parity suite, the 16-row decode-step experiments, the default bench line, its kernel trace
cd "$(dirname "$0")/.."
O=gpurun_out/r4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_call1.txt 2>&1; echo "pytest rc $?" >> $O/pytest_call1.txt
tail -n 5 $O/pytest_call1.txt
timeout 600 python tools/ar_sweep.py --out $O/ar_sweep_call1.txt --reps 2 < tools/sweeps/r04_call1.txt > $O/ar_sweep_call1.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_call1.json 2> $O/bench_call1.err; echo "bench rc $?"
cut -c1-1500 $O/bench_call1.json
PROF_OUT=$PWD/$O timeout 900 tools/prof_run.sh bench_trace python $PWD/bench.py --steps 3 --warmup 1 --no-subrecords --no-cpu-baseline --no-kernels > /dev/null 2>&1
head -n 30 $O/prof_bench_trace.txt
