#!/bin/bash
# rocprofv3 --pmc passes (counters only with --kernel-trace, one counter group per pass) -> gpurun_out/r2/pmc_r02.txt
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r2/pmc_r02.txt
mkdir -p $(dirname $OUT); : > $OUT
pass() {  # name, counters, kernel patterns (comma separated), command...
  name=$1; ctr=$2; pats=$3; shift 3
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -- "$@" > /tmp/pmc_$name.log 2>&1)
  DB=$(find /tmp/pmc_$name -name "*.db" | head -1)
  echo "### pass $name: --pmc $ctr -- $*" >> $OUT
  IFS=',' read -ra PA <<< "$pats"
  for p in "${PA[@]}"; do python $R/tools/pmc_summary.py $DB "$p" >> $OUT; done
}
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
pass sdf_fetch "FETCH_SIZE" "%sdf_query%" python $R/tools/pmc_sdf.py
pass sdf_write "WRITE_SIZE" "%sdf_query%" python $R/tools/pmc_sdf.py
pass sdf_sq "$SQ" "%sdf_query%" python $R/tools/pmc_sdf.py
pass train_sq "$SQ" "%attn_prefill_mfma%,%attn_bwd%,%attn_stats%,%sgemm_mfma%" python $R/tools/pmc_train.py
pass dec_sq "$SQ" "%attn_decode%,%dgemm%,%attn_prefill_mfma%,%sgemm_mfma%" env B=64 LC=300 STEPS=3 python $R/tools/pmc_decode.py
pass dec_fetch "FETCH_SIZE" "%attn_decode%,%dgemm%" env B=64 LC=400 STEPS=4 python $R/tools/pmc_decode.py
pass dec_write "WRITE_SIZE" "%attn_decode%,%dgemm%" env B=64 LC=400 STEPS=4 python $R/tools/pmc_decode.py
tail -n 120 $OUT
