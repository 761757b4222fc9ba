#!/bin/bash
# end-to-end A/B of decode-GEMM workgroup shapes: bench.py (2 timed passes) per SFMI_DGEMM_LDS setting; prints value + AR-loop ms/step
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
