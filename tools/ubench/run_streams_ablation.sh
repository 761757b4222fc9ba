#!/bin/bash
# what bounds the decode GEMM phase with several chains in flight?  dgemm_streams.hip (NS hipGraph chains of 24 x {qkv, proj, fc1, fc2})
# with one ingredient of dgemm_kernel removed at a time.   tools/ubench/run_streams_ablation.sh [M] [NS]
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../shapeformer_amd/csrc -DDGS_R1_ONLY"
hipcc $F dgemm_streams.hip -o /tmp/s_base 2>/dev/null &
hipcc $F -DDG_NO_MFMA dgemm_streams.hip -o /tmp/s_nomfma 2>/dev/null &
hipcc $F -DDG_NO_XLOAD dgemm_streams.hip -o /tmp/s_nox 2>/dev/null &
hipcc $F -DDG_NO_WLOAD dgemm_streams.hip -o /tmp/s_nowload 2>/dev/null &
hipcc $F -DDG_NO_XLOAD -DDG_NO_WLOAD dgemm_streams.hip -o /tmp/s_noloads 2>/dev/null &
hipcc $F -DDG_STATS_IF_LN dgemm_streams.hip -o /tmp/s_statsifln 2>/dev/null &
wait
for ns in ${2:-1 2 3 4}; do
  for v in base nomfma nox nowload noloads statsifln; do printf "%-10s " $v; /tmp/s_$v ${1:-48} $ns; done
done
