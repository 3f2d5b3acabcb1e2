"""Kernel timeline of the LAST est_pose call in a rocprofv3 kernel trace of tools/single_det.py."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
starts = [i for i, r in enumerate(rows) if "stage1_input" in r[0]]
ks = rows[starts[-1]:]
t0 = ks[0][1]
prev_end = t0
busy = 0.0
for nm, s, e in ks:
    nm = nm.replace("void p2p::", "").replace("p2p::", "").replace("(anonymous namespace)::", "").replace("pnp::", "")
    print("%-44s start %8.1f  dur %7.1f  gap %6.1f" % (nm[:44], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    busy += (e - s) / 1e3
    prev_end = max(prev_end, e)
print("kernels %d  busy %.1f us  span %.1f us" % (len(ks), busy, (prev_end - t0) / 1e3))
