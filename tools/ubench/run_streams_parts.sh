#!/bin/bash
# dgemm_kernel in throughput mode (3 chains): LayerNorm statistics / epilogue removed, with and without loads
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../shapeformer_amd/csrc -DDGS_R1_ONLY"
NL="-DDG_NO_XLOAD -DDG_NO_WLOAD"
hipcc $F dgemm_streams.hip -o /tmp/p_base 2>/dev/null &
hipcc $F -DDG_NO_STATS dgemm_streams.hip -o /tmp/p_nostats 2>/dev/null &
hipcc $F -DDG_SKIP_EPI dgemm_streams.hip -o /tmp/p_noepi 2>/dev/null &
hipcc $F -DDG_SKIP_EPI -DDG_NO_STATS dgemm_streams.hip -o /tmp/p_noepi_nostats 2>/dev/null &
hipcc $F -DDG_NO_XLOAD -DDG_NO_WLOAD -DDG_SKIP_EPI -DDG_NO_STATS dgemm_streams.hip -o /tmp/p_mfmaonly 2>/dev/null &
hipcc $F -DDG_NO_XLOAD -DDG_NO_WLOAD -DDG_NO_STATS dgemm_streams.hip -o /tmp/p_noloads_nostats 2>/dev/null &
wait
for m in ${1:-96 48}; do for v in base nostats noepi noepi_nostats noloads_nostats mfmaonly; do printf "%-16s " $v; /tmp/p_$v $m 3; done; done
