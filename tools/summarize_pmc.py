#!/usr/bin/env python3
"""Collapse the rocprofv3 CSVs of tools/profile_pmc.sh into one JSON summary for the fused kernel."""
import csv
import glob
import json
import os
import sys


def main():
    out = sys.argv[1]
    res = {"kernel": None, "counters": {}, "dispatches": {}}
    # Only the full-batch launches of the headline kernel count: the bench also launches it on single frames (latency probe),
    # and the camera-input build carries the same name stem.  "Full batch" = the largest grid of the int8-input build.
    def headline(name):
        return "yoloface56_fused" in name and "true>" not in name.replace(" ", "")
    for f in glob.glob(os.path.join(out, "trace", "*kernel_trace.csv")):
        rows = [r for r in csv.DictReader(open(f)) if headline(r["Kernel_Name"])]
        if rows:
            full = max(int(r["Grid_Size_X"]) for r in rows)
            iv3 = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", "")) for r in rows if int(r["Grid_Size_X"]) == full)
            iv = [(b, e) for b, e, _ in iv3]
            sid = [q for _, _, q in iv3]
            # bench.py alternates the timed steps between two streams: those launches OVERLAP their neighbours (a launch's ramp / drain runs beside the next
            # step's), so their own durations are longer than the step they cost.  The kernel's duration ALONE is that of the launches that overlap nobody
            # (clock settle, the kernel-time region: one stream, back to back) -- what roofline.kernel_ms of the line is compared with.
            # (an overlap counts when the neighbour runs on ANOTHER stream, or lasts more than a microsecond: back-to-back launches of ONE stream show the next
            # launch's start a few hundred nanoseconds before the previous one's end stamp -- round 6's first summary filed 728 such launches under "overlapped")
            def laps(i, j):
                return iv[j][0] < iv[i][1] and (sid[i] != sid[j] or iv[i][1] - iv[j][0] > 1000)
            lap = [(i > 0 and laps(i - 1, i)) or (i + 1 < len(iv) and laps(i, i + 1)) for i in range(len(iv))]
            alone = [e - b for (b, e), o in zip(iv, lap) if not o]
            over = [e - b for (b, e), o in zip(iv, lap) if o]
            d = alone or [e - b for b, e in iv]
            res["kernel"] = rows[0]["Kernel_Name"]
            res["trace"] = {"calls": len(d), "avg_ns": sum(d) / len(d), "min_ns": float(min(d)), "max_ns": float(max(d)), "grid_threads": full,
                            "note": "full-batch launches that overlap no other launch (the stats CSV also averages single-frame launches and the overlapped timed steps)"}
            if over:
                runs, cur = [], []
                for (b, e), o in zip(iv, lap):       # runs of consecutive overlapped launches: what a step costs there = (last end - first start) / launches
                    if o:
                        cur.append((b, e))
                    elif cur:
                        runs.append(cur); cur = []
                if cur:
                    runs.append(cur)
                span = sum(r[-1][1] - r[0][0] for r in runs)
                res["trace_overlapped"] = {"calls": len(over), "avg_duration_ns": sum(over) / len(over), "runs": len(runs), "wall_ns_per_launch": span / len(over),
                                           "note": "timed steps on two alternating streams: a launch lasts avg_duration_ns but costs wall_ns_per_launch"}
    for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            acc, cnt = {}, {}
            rows = [r for r in csv.DictReader(open(f)) if headline(r.get("Kernel_Name", ""))]
            full = max([int(r["Grid_Size"]) for r in rows] or [0])
            for r in rows:
                if int(r["Grid_Size"]) != full:
                    continue
                k = r["Counter_Name"]
                acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
                cnt[k] = cnt.get(k, 0) + 1
                for key in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
                    if key in r:
                        res["dispatches"][key] = r[key]
            for k in acc:
                res["counters"][k] = {"per_launch": acc[k] / cnt[k], "launches": cnt[k]}
    c = res["counters"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch_kb, write_kb = c["FETCH_SIZE"]["per_launch"], c["WRITE_SIZE"]["per_launch"]
        res["hbm"] = {"fetch_bytes_raw": fetch_kb * 1024, "write_bytes": write_kb * 1024,
                      "note": "FETCH_SIZE/WRITE_SIZE are in KiB; gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x "
                              "(MI355X_MICROARCH.md, HBM) -- see profiles/README.md for the calibration used"}
    # what bench.py reports as roofline.traffic / roofline_valu: stamped with the build the profile was taken on, so that
    # bench.py can refuse the numbers once the kernel sources change (bench.py: profile_counters)
    try:
        bl = json.loads(open(os.path.join(out, "bench_line.json")).read().strip().splitlines()[-1])
        cur = {"profile": os.path.basename(os.path.normpath(out)), "kernel": bl["config"]["kernel"], "source_hash": bl["config"]["kernel_source_hash"],
               "kernel_trace_avg_ns": res.get("trace", {}).get("avg_ns"), "bench_kernel_ms": bl["roofline"]["kernel_ms"]}
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if k in c:
                cur[k] = round(c[k]["per_launch"])
        if "hbm" in res:
            cur["fetch_bytes_raw"] = res["hbm"]["fetch_bytes_raw"]
            cur["write_bytes"] = res["hbm"]["write_bytes"]
            cur["hbm_bytes_per_launch"] = round(2 * res["hbm"]["fetch_bytes_raw"] + res["hbm"]["write_bytes"])   # gfx950: FETCH_SIZE counts half of a streaming read
        json.dump(cur, open(os.path.join(out, "pmc_current.json"), "w"), indent=1)
    except Exception as e:      # noqa: BLE001
        print("pmc_current.json not written:", e)
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main()
