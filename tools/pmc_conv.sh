#!/bin/bash
# rocprofv3 --pmc passes over tools/kbench_conv.py (UNet3D + up-sampler at 64 shapes): MFMA busy cycles and HBM-side bytes of the
# conv3d_igemm_kernel instances (one counter group per pass, --kernel-trace only) -> gpurun_out/r5/pmc_conv.txt (or $PMC_OUT)
export TMPDIR=/tmp
R=$PWD
OUT=${PMC_OUT:-$R/gpurun_out/r5/pmc_conv.txt}
mkdir -p $(dirname $OUT); : > $OUT
pass() {
  name=$1; ctr=$2; shift 2
  rm -rf /tmp/pmc_$name
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_$name -- "$@" > /tmp/pmc_$name.log 2>&1)
  DB=$(find /tmp/pmc_$name -name "*.db" | head -1)
  echo "### pass $name: --pmc $ctr -- $*" >> $OUT
  python $R/tools/pmc_summary.py $DB "%conv3d_igemm%" >> $OUT
}
pass conv_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" python $R/tools/kbench_conv.py --reps 2
pass conv_fetch "FETCH_SIZE" python $R/tools/kbench_conv.py --reps 2
pass conv_write "WRITE_SIZE" python $R/tools/kbench_conv.py --reps 2
cat $OUT
