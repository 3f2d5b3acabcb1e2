"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max / share.

    python tools/rocprof_summary.py gpurun_out/prof_r01/bench_results.db > profiles/r01_bench_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by %s order by sum(end-start) desc" % (name_col, name_col)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# total kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[1] for r in rows)))
    print("%-110s %8s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
    for n, c, t, a, mn, mx in rows:
        print("%-110s %8d %12.3f %12.2f %12.2f %12.2f %6.2f%%" % (n[:110], c, t / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total))


if __name__ == "__main__":
    main(sys.argv[1])
