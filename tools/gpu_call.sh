#!/bin/bash
# One parametrised driver for a GPU-box call (replaces the per-call scripts of rounds 3-4).  Each argument is a step, run in order,
# every step under its own timeout, logs under gpurun_out/<tag>/ (copy what should be kept into profiles/):
#
#   tools/gpu_call.sh <tag> step [step ...]
#
#   pytest[=<pytest args>]        python -m pytest <args, default "tests -m gpu -x -q">        -> pytest_<n>.txt
#   py=<script and args>          python <script and args>                                       -> py_<n>.txt
#   sweep=<tools/sweeps file>[,reps[,steps]]   tools/ar_sweep.py < file                           -> ar_sweep_<file>.txt
#   bench[=<bench.py args>]       python bench.py <args, default the driver's "--gpus 1 --steps 20 --warmup 5">  -> bench_<n>.json
#   trace=<name>:<command>        rocprofv3 --kernel-trace --stats of <command> (tools/prof_run.sh)  -> prof_<name>.txt
#   pmc=<name>:<counters>:<kernel LIKE pattern>:<command>   one rocprofv3 --pmc pass (--kernel-trace only, as the pool requires) and the
#                                 per-kernel counter averages (tools/pmc_summary.py); counters separated by "+"     -> pmc_<name>.txt
# Steps that contain spaces must be quoted by the caller.
cd "$(dirname "$0")/.."
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
n=0
for step in "$@"; do
  n=$((n + 1))
  kind=${step%%=*}; arg=""; [[ "$step" == *=* ]] && arg=${step#*=}
  echo "=== [$n] $step"
  case $kind in
    pytest)
      timeout 1500 python -m pytest ${arg:-tests -m gpu -x -q} > $O/pytest_$n.txt 2>&1; echo "pytest rc $?" >> $O/pytest_$n.txt; tail -n 6 $O/pytest_$n.txt ;;
    ubench)      # ubench=<name>: compile tools/ubench/<name>.hip and run it                      -> ubench_<name>.txt
      hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/$arg.hip -o /tmp/ub_$arg > $O/ubench_$arg.txt 2>/dev/null; timeout 300 /tmp/ub_$arg >> $O/ubench_$arg.txt 2>&1; echo "rc $?" >> $O/ubench_$arg.txt; tail -n 20 $O/ubench_$arg.txt ;;
    py)
      timeout 900 python $arg > $O/py_$n.txt 2>&1; echo "rc $?" >> $O/py_$n.txt; tail -n 40 $O/py_$n.txt ;;
    sweep)
      IFS=, read -r f reps steps <<< "$arg"; reps=${reps:-2}; steps=${steps:-512}
      b=$(basename $f .txt)
      timeout 900 python tools/ar_sweep.py --out $O/ar_sweep_$b.txt --reps $reps --steps $steps < $f > $O/ar_sweep_$b.log 2>&1; echo "rc $?"; tail -n 12 $O/ar_sweep_$b.txt ;;
    bench)
      timeout 1200 python bench.py ${arg:---gpus 1 --steps 20 --warmup 5} > $O/bench_$n.json 2> $O/bench_$n.err; echo "bench rc $?"; cut -c1-600 $O/bench_$n.json; tail -n 3 $O/bench_$n.err ;;
    trace)
      name=${arg%%:*}; cmd=${arg#*:}
      PROF_OUT=$PWD/$O timeout 1000 tools/prof_run.sh $name $cmd > $O/trace_$name.log 2>&1; head -n 30 $O/prof_$name.txt | cut -c1-200 ;;
    pmc)
      name=${arg%%:*}; rest=${arg#*:}; ctr=${rest%%:*}; rest=${rest#*:}; pat=${rest%%:*}; cmd=${rest#*:}
      rm -rf /tmp/pmc_$name
      (cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --pmc ${ctr//+/ } -d /tmp/pmc_$name -- $cmd > /tmp/pmc_$name.log 2>&1)
      DB=$(find /tmp/pmc_$name -name "*.db" | head -1)
      { echo "### --pmc ${ctr//+/ } -- $cmd"; python tools/pmc_summary.py $DB "$pat"; } > $O/pmc_$name.txt 2>&1; tail -n 40 $O/pmc_$name.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
