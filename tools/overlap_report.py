"""Stream-mode kernel trace against a blocking one: per kernel (name, grid) the average duration in both, and what the difference
sums to per step -- where the overlapped PnP tail costs the generator kernels time.
    python tools/overlap_report.py <stream_results.db> <blocking_results.db> <steps_stream> <steps_blocking>"""
import sqlite3
import sys
from collections import defaultdict


def load(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
    q = "select name, start, end%s from kernels order by start" % ((", " + gx) if gx else "")
    return db.execute(q).fetchall()


def short(n):
    return n.replace("void p2p::", "").replace("p2p::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]


def main(sp, bp, ss, sb):
    S, B = load(sp), load(bp)
    def agg(rows):
        d = defaultdict(lambda: [0, 0.0])
        for r in rows:
            k = (short(r[0]), r[3] if len(r) > 3 else 0)
            d[k][0] += 1; d[k][1] += (r[2] - r[1]) / 1e3
        return d
    a, b = agg(S), agg(B)
    tot = 0.0
    rows = []
    for k in a:
        if k not in b:
            continue
        avs, avb = a[k][1] / a[k][0], b[k][1] / b[k][0]
        per_step = (avs - avb) * a[k][0] / ss
        tot += per_step
        rows.append((per_step, k, a[k][0] / ss, avs, avb))
    rows.sort(reverse=True)
    print("%-62s %9s %8s %10s %10s %12s" % ("kernel", "grid", "n/step", "stream us", "block us", "d us/step"))
    for per_step, k, n, avs, avb in rows[:30]:
        print("%-62s %9d %8.1f %10.1f %10.1f %12.1f" % (k[0], k[1], n, avs, avb, per_step))
    print("sum of differences: %.1f us per step" % tot)
    # GPU idle (no kernel running) and span per step in the stream trace
    ev = sorted((r[1], r[2]) for r in S)
    t0, t1 = ev[0][0], max(e for _, e in ev)
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("stream trace: span %.1f ms, some kernel running %.1f ms, idle %.1f ms (%d steps incl. warm-up)" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, ss))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4]))
