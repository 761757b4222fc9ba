#!/bin/bash
# Device ISA of every csrc/*.hip for gfx950: per kernel scratch_ instruction count, VGPR / AGPR / SGPR use, LDS, occupancy.
#   tools/isa_check.sh [outdir]        (no GPU needed; ~1 min)
# A launched kernel with scratch_ instructions spills registers to memory: fix it.
cd "$(dirname "$0")/.."
OUT=${1:-/tmp/isa}
mkdir -p "$OUT"
for f in shapeformer_amd/csrc/*.hip; do
  b=$(basename "$f" .hip)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Iinclude -Ishapeformer_amd/csrc --cuda-device-only -S "$f" -o "$OUT/$b.s" 2>/dev/null &
done
wait
python3 - "$OUT" <<'PY'
import glob, re, subprocess, sys
tot = 0
for f in sorted(glob.glob(sys.argv[1] + "/*.s")):
    name, scr, meta = None, {}, {}
    for ln in open(f):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith(".") and not m.group(1).startswith("BB"):
            name = m.group(1)
        if name and "scratch_" in ln and not ln.lstrip().startswith(";"):
            scr[name] = scr.get(name, 0) + 1
        m = re.match(r"^\s*;\s*(NumVgprs|NumAgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize):\s*(\d+)", ln)
        if m and name:
            meta.setdefault(name, {})[m.group(1)] = int(m.group(2))
    for k, v in meta.items():
        if scr.get(k) or v.get("ScratchSize"):
            tot += 1
            dm = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            print(f"SPILL {f.split('/')[-1]:18s} {dm[:110]:110s} scratch_insts={scr.get(k,0)} {v}")
print(f"kernels with scratch: {tot}")
PY
