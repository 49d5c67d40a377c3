"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / % -- the same table
`rocprofv3 --stats` prints, usable when only the .db was produced."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(rocpd_kernel_dispatch)")]
rows = cur.execute("""select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start)
                      from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                      group by s.kernel_name order by 3 desc""").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':70s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'%':>6s}")
for name, n, t, a, mn, mx in rows:
    print(f"{name[:70]:70s} {n:8d} {t/1e6:10.2f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:10.2f} {100*t/tot:6.2f}")
print(f"total kernel time {tot/1e6:.2f} ms")
