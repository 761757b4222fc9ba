#!/bin/bash
# The AR loop (4 x 96 rows, turnstile) under HIP-runtime environment knobs that touch graph launch / dispatch / signalling.
out=gpurun_out/r3/ar_sweep_env.txt; mkdir -p gpurun_out/r3; : > $out
for e in "X=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "HIP_FORCE_DEV_KERNARG=0" "HIP_FORCE_DEV_KERNARG=1" \
         "ROC_SYSTEM_SCOPE_SIGNAL=0" "DEBUG_HIP_DYNAMIC_QUEUES=0" "DEBUG_HIP_DYNAMIC_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "GPU_STREAMOPS_CP_WAIT=1" \
         "DEBUG_HIP_KERNARG_COPY_OPT=0" "ROC_ACTIVE_WAIT_TIMEOUT=0" "X=1"; do
  echo "### $e" >> $out
  env $e timeout 200 python tools/ar_sweep.py --out $out < tools/sweeps/r03_env.txt > /dev/null 2>&1 || echo "   (failed)" >> $out
done
grep -a "###\|ms/step\|failed" $out | cut -c1-24,70-170
