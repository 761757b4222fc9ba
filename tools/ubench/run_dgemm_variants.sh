#!/bin/bash
# decode-GEMM tuning variants (waves per workgroup, loads in flight) in the C++ graph chain; arg: rows (default 64)
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../shapeformer_amd/csrc"
hipcc $F dgemm_chain.hip -o /tmp/d_base &
hipcc $F -DDG_FORCE_UN=4 dgemm_chain.hip -o /tmp/d_un4 &
hipcc $F -DDG_FORCE_NW=16 -DDG_FORCE_UN=1 dgemm_chain.hip -o /tmp/d_nw16un1 &
hipcc $F -DDG_FORCE_NW=16 -DDG_FORCE_UN=2 dgemm_chain.hip -o /tmp/d_nw16un2 &
hipcc $F -DDG_FORCE_UN=1 dgemm_chain.hip -o /tmp/d_un1 &
wait
for v in base un4 un1 nw16un1 nw16un2; do echo "== $v"; timeout 60 /tmp/d_$v ${1:-64} | grep -v "S1\|S2\|head\|plain"; done
