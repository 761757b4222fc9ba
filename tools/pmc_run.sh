#!/bin/bash
# rocprofv3 --pmc passes (counters only with --kernel-trace, one group per pass) -> gpurun_out/pmc_r01b.txt
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
: > gpurun_out/pmc_r01b.txt
pass() {  # name, counters, command...
  name=$1; ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -- "$@" > /tmp/pmc_$name.log 2>&1)
  DB=$(find /tmp/pmc_$name -name "*.db" | head -1)
  echo "### pass $name: --pmc $ctr -- $*" >> gpurun_out/pmc_r01b.txt
  python tools/pmc_summary.py $DB "%sdf_query%" >> gpurun_out/pmc_r01b.txt
  python tools/pmc_summary.py $DB "%attn_%" >> gpurun_out/pmc_r01b.txt
  python tools/pmc_summary.py $DB "%dgemm%" >> gpurun_out/pmc_r01b.txt
}
pass sdf_fetch "FETCH_SIZE" python $R/tools/pmc_sdf.py
pass sdf_write "WRITE_SIZE" python $R/tools/pmc_sdf.py
pass sdf_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python $R/tools/pmc_sdf.py
pass dec_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" env B=64 LC=300 STEPS=3 python $R/tools/pmc_decode.py
tail -n 80 gpurun_out/pmc_r01b.txt
