#!/bin/bash
# builds the ablation variants of the wide decode GEMM on the GPU box and times them
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../shapeformer_amd/csrc"
hipcc $F dgemm_wide_chain.hip -o /tmp/w_base &
hipcc $F '-DXIDX(i)=0' dgemm_wide_chain.hip -o /tmp/w_nox &
hipcc $F '-DDG_MFMA(a,b,c)=(c)' dgemm_wide_chain.hip -o /tmp/w_nomfma &
hipcc $F -DWIDE_SKIP_EPI dgemm_wide_chain.hip -o /tmp/w_noepi &
wait
for v in base nox nomfma noepi; do echo "== $v"; /tmp/w_$v ${1:-192}; done
