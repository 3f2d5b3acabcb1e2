"""Per-kernel SQ counter summary from rocprofv3 --pmc passes (one or more rocpd databases): counters are summed over a
kernel's launches and reported per launch, plus unit-free ratios between counters of the same block.

    python tools/pmc_sq.py gpurun_out/pmc_sq1/x_results.db gpurun_out/pmc_sq2/x_results.db > profiles/r01_sq_counters.txt
"""
import sqlite3
import sys


def main(paths):
    match = None
    if "--match" in paths:              # only kernels whose name contains this substring
        i = paths.index("--match")
        match = paths[i + 1]
        paths = paths[:i] + paths[i + 2:]
    agg = {}
    for path in paths:
        db = sqlite3.connect(path)
        for name, counter, n, total, dur in db.execute("select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection group by kernel_name, counter_name"):
            agg.setdefault(name, {})[counter] = (n, total)
            agg[name]["_duration_ns_" + counter] = (n, dur)
    keys = sorted((k for k in agg if match is None or match in k), key=lambda k: -agg[k].get("SQ_BUSY_CYCLES", (0, 0))[1])
    print("# rocprofv3 --pmc, bench.py --blocking --steps 1 --warmup 1 (two warm-up + timed + two profiling steps); values per launch, summed over the chip")
    for k in keys[:8]:
        c = {n: v[1] / max(v[0], 1) for n, v in agg[k].items()}
        print("\n%s   (%d launches)" % (k.replace("void p2p::", "").replace("(anonymous namespace)::", "")[:90], max(v[0] for v in agg[k].values())))
        for n in sorted(c):
            if not n.startswith("_"):
                print("    %-34s %16.0f" % (n, c[n]))
        wc = c.get("SQ_WAVE_CYCLES", 0)
        if wc:
            print("    -- of wave cycles: issuing %.1f %%, issue-stalled %.1f %% (on LDS %.1f %%), parked in s_waitcnt / barrier %.1f %%" % (
                100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * c.get("SQ_WAIT_INST_LDS", 0) / wc,
                100 * c.get("SQ_WAIT_ANY", 0) / wc))
        if c.get("SQ_LDS_IDX_ACTIVE"):
            print("    -- LDS bank-conflict cycles / LDS active cycles: %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES is per SIMD, summed over the chip's 1024 SIMDs (= 32 cycles x number of 32x32x16 MFMAs: checked
            # against the algorithmic count); GRBM_GUI_ACTIVE is summed over the 8 XCDs
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0
            print("    -- MFMA pipe busy: %.1f %% of SIMD cycles; shader clock during the (profiled) launches: %.2f GHz" % (
                100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), cyc / c["_duration_ns_GRBM_GUI_ACTIVE"]))


if __name__ == "__main__":
    main(sys.argv[1:])
