#!/usr/bin/env python3
"""Summarise rocprofv3 output of `bench.py` into the small JSON / CSV files kept under profiles/.

    python tools/pmc_summary.py stats  <dir-of-kernel-trace-run>  <steps+warmup>  out.csv
    python tools/pmc_summary.py pmc    <dir-of-FETCH_SIZE-run> <dir-of-WRITE_SIZE-run> <steps+warmup> out.json

The PMC passes are collected separately (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass) and corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB; FETCH_SIZE reports
half of the bytes of wide (16 B/lane) streaming reads, so it is doubled; WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\.kd$", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def source_sha16():
    """the kernel sources this profile was taken with (bench.py compares it with the sources it runs)"""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.source_sha16()


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    if not hits:
        raise SystemExit(f"no {pat} under {d}")
    return hits[0]


def stats(d, nsteps, out):
    rows = defaultdict(lambda: [0, 0.0])
    with open(find(d, "*kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            rows[k][0] += 1
            rows[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(v[1] for v in rows.values())
    with open(out, "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches_per_step", "avg_us", "ms_per_step", "percent"])
        for k, (c, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, round(c / nsteps, 2), round(us / c, 2), round(us / nsteps / 1e3, 4), round(100 * us / tot, 2)])
        w.writerow(["TOTAL", "", "", round(tot / nsteps / 1e3, 4), 100.0])


def counter(d, name):
    acc = defaultdict(lambda: [0, 0.0])
    with open(find(d, "*counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != name:
                continue
            k = short(r["Kernel_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return acc


def pmc(dfetch, dwrite, nsteps, out):
    fe, wr = counter(dfetch, "FETCH_SIZE"), counter(dwrite, "WRITE_SIZE")
    res = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py; KiB -> MB; "
                    "FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md); averages per launch",
           "_source_sha16": source_sha16()}
    for k in sorted(fe, key=lambda k: -fe[k][1]):
        c, kib = fe[k]
        wc, wkib = wr.get(k, (0, 0.0))
        res[k] = {"launches_per_step": round(c / nsteps, 2),
                  "fetch_MB_per_launch_corrected": round(2 * kib * 1024 / c / 1e6, 2),
                  "write_MB_per_launch": round(wkib * 1024 / wc / 1e6, 2) if wc else None}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


def mfma(d, nsteps, out):
    """Matrix-pipe occupancy per kernel from one PMC pass (GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES) joined
    with the kernel durations of the same pass.  MI355X_MICROARCH.md: GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the
    clock a dispatch held is GRBM_GUI_ACTIVE / 8 / duration (reads high on dispatches under ~0.3 ms);
    SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over the chip's 1024 SIMDs (256 CUs x 4), so
    busy / (1024 * GRBM_GUI_ACTIVE / 8) is the fraction of SIMD-cycles with an MFMA in flight."""
    vals = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    dur = defaultdict(float)
    seen = set()
    with open(find(d, "*counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            vals[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
                if r.get("End_Timestamp") and r.get("Start_Timestamp"):
                    dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    res = {"_note": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES over bench.py; "
                    "per kernel: launches/step, clock_GHz = GRBM_GUI_ACTIVE/8/duration, mfma_busy_frac = "
                    "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8)"}
    for k in sorted(vals, key=lambda k: -vals[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
        v = vals[k]
        gui = v.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if gui <= 0:
            continue
        ent = {"launches_per_step": round(cnt[k] / nsteps, 2),
               "mfma_busy_frac": round(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * gui), 4)}
        if dur[k] > 0:
            ent["clock_GHz"] = round(gui / dur[k] / 1e3, 3)
            ent["avg_us"] = round(dur[k] / cnt[k], 2)
        res[k] = ent
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


def timeline(d, nsteps, out):
    """every launch of the LAST step in stream order: start offset, duration, gap to the previous kernel, grid, kernel"""
    with open(find(d, "*kernel_trace.csv")) as f:
        rows = sorted(csv.DictReader(f), key=lambda r: int(r["Start_Timestamp"]))
    per = len(rows) // nsteps
    last = rows[len(rows) - per:]
    # align on the first stem kernel of the step if the split is off by the probe step's extra launches
    t0 = int(last[0]["Start_Timestamp"])
    prev_end = t0
    with open(out, "w") as f:
        f.write("# launches of the last step (rocprofv3 --kernel-trace): start_us dur_us gap_us grid kernel\n")
        for r in last:
            s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
            f.write("%9.1f %8.1f %6.1f %8s  %s\n" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, (s_ - prev_end) / 1e3, grid,
                                                  short(r["Kernel_Name"])[:110]))
            prev_end = e_
        f.write("# span %.1f us, kernel time %.1f us\n" % ((prev_end - t0) / 1e3,
                                                           sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "timeline":
        timeline(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5])
