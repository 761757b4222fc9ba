"""Concurrency of the two decode kernel families from a rocprofv3 --kernel-trace rocpd database: how much of the wall time has an
attention launch AND a decode-GEMM launch (of another chain) in flight, and how long a GEMM launch takes with / without a KV
stream beside it.   python tools/overlap_stats.py <db> [skip_fraction]"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5       # ignore the first part of the trace (prefill, graph capture, warm-up)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
rows = None
for t in tabs:
    cols = [r[1] for r in db.execute(f"pragma table_info('{t}')")]
    low = [c.lower() for c in cols]
    name = next((c for c in cols if c.lower() in ("name", "kernel_name", "kernel")), None)
    if name and "start" in low and "end" in low and "kernel" in t.lower():
        try:
            rows = list(db.execute(f"select {name}, start, end from '{t}' where end > start"))
        except Exception:
            rows = None
        if rows:
            print(f"# dispatches from '{t}' ({len(rows)} rows)")
            break
if not rows:
    print("no (name, start, end) dispatch table found; tables:", tabs)
    for t in tabs:
        print(t, [r[1] for r in db.execute(f"pragma table_info('{t}')")][:12])
    sys.exit(1)
rows.sort(key=lambda r: r[1])
t0, t1 = rows[0][1], max(r[2] for r in rows)
lo = t0 + (t1 - t0) * skip
cls = lambda n: "attn" if "attn_decode" in n else ("gemm" if "dgemm_kernel" in n else "other")
ev = []
sel = [(cls(n), s, e) for n, s, e in rows if s >= lo and cls(n) != "other"]
for c, s, e in sel:
    ev.append((s, 1, c)); ev.append((e, -1, c))
ev.sort()
act = {"attn": 0, "gemm": 0}
last = ev[0][0]
tm = {"attn_only": 0, "gemm_only": 0, "both": 0, "idle": 0}
hist = {}
for t, d, c in ev:
    dt = t - last
    a, g = act["attn"], act["gemm"]
    tm["both" if a and g else "attn_only" if a else "gemm_only" if g else "idle"] += dt
    hist[(a, g)] = hist.get((a, g), 0) + dt
    act[c] += d
    last = t
wall = sum(tm.values())
print(f"window {wall / 1e6:.2f} ms (last {100 * (1 - skip):.0f}% of the trace), {len(sel)} attention / GEMM dispatches")
for k, v in tm.items():
    print(f"  {k:10s} {100 * v / wall:5.1f} %")
print("  (attention launches in flight, GEMM launches in flight) -> % of the window:")
for k in sorted(hist):
    if hist[k] / wall > 0.005:
        print(f"     {k}: {100 * hist[k] / wall:5.1f}")
# GEMM launch duration by whether an attention launch overlapped it
import bisect
att = sorted((s, e) for c, s, e in sel if c == "attn")
starts = [s for s, _ in att]
def overlapped(s, e):
    i = bisect.bisect_left(starts, e)
    return any(att[j][1] > s for j in range(max(0, i - 8), i))
dur = {True: [], False: []}
for c, s, e in sel:
    if c == "gemm":
        dur[overlapped(s, e)].append(e - s)
for k in (False, True):
    if dur[k]:
        d = sorted(dur[k])
        print(f"  GEMM launches {'beside' if k else 'without'} an attention launch: {len(d):6d}, mean {sum(d) / len(d) / 1e3:6.2f} us, median {d[len(d) // 2] / 1e3:6.2f} us")
ad = sorted(e - s for c, s, e in sel if c == "attn")
print(f"  attention launches: {len(ad)}, mean {sum(ad) / len(ad) / 1e3:.2f} us")
