"""HBM traffic of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in
SEPARATE runs, as MI355X_MICROARCH.md prescribes: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2).

    python tools/pmc_traffic.py gpurun_out/pmc_fetch/x_results.db gpurun_out/pmc_write/x_results.db profiles/r01_traffic.json

Units/corrections (MI355X_MICROARCH.md section HBM): the counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read -> doubled here; WRITE_SIZE is
used as is (uncalibrated)."""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), sum(value) from counters_collection where counter_name=? group by kernel_name",
                      (counter,)).fetchall()
    return {r[0]: (r[1], r[2]) for r in rows}


def main(fetch_db, write_db, out):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    res = {}
    for k in f:
        n, fs = f[k]
        nw, ws = w.get(k, (n, 0.0))
        res[k] = {"launches": n, "fetch_bytes_per_launch": 2.0 * fs * 1024 / n, "write_bytes_per_launch": ws * 1024 / max(nw, 1),
                  "hbm_bytes_per_launch": (2.0 * fs * 1024 / n) + ws * 1024 / max(nw, 1),
                  "note": "FETCH_SIZE doubled (gfx950 half-count on wide coalesced reads); WRITE_SIZE uncalibrated"}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:8]:
        print("%-60s launches %5d  fetch %.1f MB  write %.1f MB per launch" % (k[:60], v["launches"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:4])
