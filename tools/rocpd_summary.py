"""Prints per-kernel time statistics and PMC counter sums from rocprofv3 rocpd (.db) outputs."""
import sqlite3
import sys


def main(paths):
    for path in paths:
        con = sqlite3.connect(path)
        cur = con.cursor()
        print(f"== {path}")
        try:
            rows = cur.execute(
                "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"
            ).fetchall()
            print(f"{'kernel':70s} {'calls':>6s} {'total_ns':>12s} {'avg_ns':>12s} {'min_ns':>12s} {'max_ns':>12s}")
            for r in rows:
                print(f"{r[0][:70]:70s} {r[1]:6d} {r[2]:12d} {r[3]:12.0f} {r[4]:12d} {r[5]:12d}")
        except sqlite3.Error as e:
            print("kernels view:", e)
        try:
            cols = [c[1] for c in cur.execute("pragma table_info(counters_collection)")]
            if cols:
                name_col = "kernel_name" if "kernel_name" in cols else "name"
                rows = cur.execute(
                    f"select {name_col}, counter_name, count(*), sum(value) from counters_collection group by 1, 2 order by 1, 2"
                ).fetchall()
                for r in rows:
                    print(f"  {r[0][:60]:60s} {r[1]:28s} dispatches={r[2]:4d} sum={r[3]:.6g} per_dispatch={r[3] / r[2]:.6g}")
        except sqlite3.Error as e:
            print("counters_collection view:", e)


if __name__ == "__main__":
    main(sys.argv[1:])
