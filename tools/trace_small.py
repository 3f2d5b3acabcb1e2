"""Per-kernel durations of ONE generator pass at a small batch size from a rocprofv3 rocpd database (the last pass in the trace).
Usage: python tools/trace_small.py <results.db> [n_kernels_per_pass]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    # a pass starts at the first-layer kernel
    starts = [i for i, r in enumerate(rows) if "conv1_f16x3" in r[0] or "conv_first" in r[0]]
    a = starts[-1]
    ks = rows[a:]
    t0 = ks[0][1]
    tot = 0.0
    for nm, s, e in ks:
        nm = nm.replace("void p2p::", "").replace("p2p::", "").replace("(anonymous namespace)::", "")
        print("%-46s start %8.1f us  dur %7.1f us" % (nm[:46], (s - t0) / 1e3, (e - s) / 1e3))
        tot += (e - s) / 1e3
    print("kernels %d  sum %.1f us  span %.1f us" % (len(ks), tot, (ks[-1][2] - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
