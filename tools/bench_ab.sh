#!/bin/bash
# A/B of bench.py settings on ONE box: every argument is one quoted flag string, e.g. tools/bench_ab.sh "--lanes 0" "--lanes 2"
mkdir -p gpurun_out/r3
for flags in "$@"; do
  timeout 300 python bench.py $flags --steps 3 --warmup 1 --no-cpu-baseline --no-kernels --no-subrecords 2>/dev/null > /tmp/ab.json
  python - "$flags" <<'PY'
import json, sys
for ln in open("/tmp/ab.json"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(sys.argv[1], "| value", d["value"], "ms", d["ms_per_step"], d.get("stages_ms"), "ar ms/step", d.get("ar_loop", {}).get("ms_per_step"),
              d.get("turnstile"), "probe", d.get("chain_stream_probe_ms"), flush=True)
PY
done
