#!/bin/bash
# end-to-end A/B of decode-step settings: bench.py (2 timed passes) per stdin line `name _ [WIDE] [SKIP=<spec>] [bench args]`;
# WIDE = two-n-tiles-per-wave decode GEMM for 17..96 rows, SKIP = SFMI_DECODE_SKIP timing ablation (gemm | attn | gemm@0,attn@1,...);
# prints value + stage times + AR-loop ms/step.  (profiles/r02_sweep_*.txt)
out=${1:-gpurun_out/r2/sweep.txt}; mkdir -p $(dirname $out); : > $out
run() {  # name, env, extra args
  echo "== $1 [$2] $3" >> $out
  local sk=""; local ex="$3"
  local wide=0
  if [[ "$ex" == WIDE* ]]; then wide=1; ex="${ex#WIDE}"; fi
  if [[ "$ex" == SKIP=* ]]; then sk="${ex%% *}"; sk="${sk#SKIP=}"; ex="${ex#SKIP=$sk}"; fi
  SFMI_DGEMM_WIDE=$wide SFMI_DECODE_SKIP="$sk" timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernels $ex 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('   value', d['value'], 'ms/step', d['ms_per_step'], 'stages', d.get('stages_ms'), 'ar ms/step', d.get('ar_loop', {}).get('ms_per_step'))
" >> $out
}
while read -r name env extra; do [ -n "$name" ] && run "$name" "${env//_/}" "$extra"; done
