"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd database."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "%"
rows = list(c.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like ? group by kernel_name, counter_name order by kernel_name, counter_name", (pat,)))
cur = None
for k, cn, n, v in rows:
    if k != cur:
        print(f"\n{k[:90]}  ({n} dispatches)")
        cur = k
    print(f"   {cn:<28}{v:>16.1f}")
