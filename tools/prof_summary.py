"""Summarise a rocprofv3 rocpd database (kernel-trace) into a small text table (for profiles/)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print(f"# total kernel time {tot/1e3:.2f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'kernel':<72}{'calls':>8}{'total_ms':>11}{'avg_us':>10}{'pct':>7}")
for n, k, t, a, p in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{n[:70]:<72}{k:>8}{t/1e3:>11.2f}{a:>10.2f}{p:>7.2f}")
