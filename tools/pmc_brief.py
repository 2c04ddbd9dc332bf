"""Brief per-kernel PMC table from rocprofv3 rocpd databases: python tools/pmc_brief.py kernel-substring db..."""
import sqlite3
import sys

needle = sys.argv[1]
for path in sys.argv[2:]:
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = cur.execute(
        f"select {name_col}, counter_name, count(*), sum(value) from counters_collection where {name_col} like ? group by 1, 2 order by 1, 2",
        (f"%{needle}%",),
    ).fetchall()
    k = cur.execute("select name, count(*), avg(end-start) from kernels where name like ? group by 1", (f"%{needle}%",)).fetchall()
    for r in k:
        print(f"{path}: {r[0][:80]} calls={r[1]} avg_ns={r[2]:.0f}")
    for r in rows:
        print(f"    {r[1]:28s} per_dispatch={r[3] / r[2]:.6g}")
