#!/bin/bash
# Per-kernel launch durations (rocprofv3 --kernel-trace --stats keeps the chains concurrent; --pmc would serialise the dispatches)
# of ONE attention-only chain, ONE GEMM-only chain and both together (80 rows each, 128 steps from cached lengths 300..416 = the mid-run lengths of the bench): how much longer is a decode-GEMM
# launch while a KV stream saturates HBM?  -> gpurun_out/r3/prof_overlap.txt (kept as profiles/r03_prof_overlap.txt)
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r3/prof_overlap.txt
mkdir -p $(dirname $OUT); : > $OUT
run() {  # name, config line
  rm -rf /tmp/po_$1
  echo "$2" > /tmp/po_$1.cfg
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/po_$1 -- python $R/tools/ar_sweep.py --steps 128 --out /tmp/po_$1.txt < /tmp/po_$1.cfg > /tmp/po_$1.log 2>&1)
  DB=$(find /tmp/po_$1 -name "*.db" | head -1)
  { echo "## $2"; grep -a "ms/step" /tmp/po_$1.txt | cut -c1-200; python $R/tools/prof_summary.py $DB 6 | grep -a "kernel\|dgemm_kernel\|attn_decode\|sample_kernel\|attn_gate";
    python $R/tools/overlap_stats.py $DB 0.6; echo; } >> $OUT
}
run attn  "attn_alone rows=160 chains=2 lanes=0 lclo=300 lchi=416 ablate=gemm@0,gemm@1,attn@1"
run gemm  "gemm_alone rows=160 chains=2 lanes=0 lclo=300 lchi=416 ablate=gemm@0,attn@0,attn@1"
run both  "together   rows=160 chains=2 lanes=0 lclo=300 lchi=416 ablate=gemm@0,attn@1"
run real4 "real_4x80_lanes0 rows=320 chains=4 lanes=0 lclo=300 lchi=416"
run real4t "real_4x80_lanes2 rows=320 chains=4 lanes=2 lclo=300 lchi=416"
run real4b "real_4x96_lanes2 rows=384 chains=4 lanes=2 lclo=300 lchi=416"
cat $OUT
