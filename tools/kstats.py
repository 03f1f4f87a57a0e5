"""Diagnostic: per-kernel statistics (count, average / min / max duration in us) from a rocprofv3 rocpd database (--kernel-trace).
usage: python tools/kstats.py gpurun_out/x/prof/name_results.db [> profiles/xxx_kernel_stats.csv]"""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from {disp} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 6 desc"
rows = list(cur.execute(q))
tot = sum(r[5] for r in rows)
print('"Name","Calls","TotalDurationUs","AverageUs","MinUs","MaxUs","Percentage"')
for n, c, a, mn, mx, sm in rows:
    print(f'"{n}",{c},{sm:.3f},{a:.3f},{mn:.3f},{mx:.3f},{100*sm/tot:.2f}')
