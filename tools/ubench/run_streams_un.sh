#!/bin/bash
# decode GEMM phase at M rows with UN = 1 / 2 k16-steps of loads in flight (dgemm_streams.hip, 3 chains), + the no-load / no-MFMA ablations
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../shapeformer_amd/csrc -DDGS_R1_ONLY"
hipcc $F dgemm_streams.hip -o /tmp/u_base 2>/dev/null &
hipcc $F -DDG_FORCE_UN=2 dgemm_streams.hip -o /tmp/u_un2 2>/dev/null &
hipcc $F -DDG_FORCE_UN=2 -DDG_NO_XLOAD -DDG_NO_WLOAD dgemm_streams.hip -o /tmp/u_un2_noloads 2>/dev/null &
hipcc $F -DDG_FORCE_UN=2 -DDG_NO_MFMA dgemm_streams.hip -o /tmp/u_un2_nomfma 2>/dev/null &
wait
for m in ${1:-96 80}; do for v in base un2 un2_noloads un2_nomfma; do printf "%-12s " $v; /tmp/u_$v $m 3; done; done
