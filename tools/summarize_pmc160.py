#!/usr/bin/env python3
"""SUPERSEDED by tools/summarize_secondary.py (round 5).  Collapse the rocprofv3 CSVs of the round 2-4 form of tools/profile_160.sh into a per-kernel table (160x160 path: band_k1..k4 or the 27 stage kernels)."""
import csv
import glob
import json
import os
import sys


def short(name):
    for k in ("band_k1", "band_k23", "band_k2", "band_k3", "band_k4"):
        if k in name:
            return k
    if "generic_stage_kernel" in name:
        return "stage_" + name.split("generic_stage_kernelILi")[1].split("E")[0] if "ILi" in name else "stage"
    return None


def main():
    out = sys.argv[1]
    res = {}
    # kernel time of the FULL-BATCH launches (1024 frames): the command also runs a 6-frame parity batch, which the --stats
    # average would mix in
    for f in glob.glob(os.path.join(out, "trace", "*kernel_trace.csv")):
        per = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and int(r["Grid_Size_X"]) >= 256 * 512:
                per.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in per.items():
            res.setdefault(k, {})["trace"] = {"calls": len(v), "avg_us": sum(v) / len(v), "min_us": min(v)}
    for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
            acc = {}
            for r in csv.DictReader(open(f)):
                k = short(r.get("Kernel_Name", ""))
                if not k or int(r["Grid_Size"]) < 256 * 512:        # only the full-batch launches (1024 frames)
                    continue
                acc.setdefault((k, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
                res.setdefault(k, {})["vgprs"] = r.get("VGPR_Count")
                res[k]["lds"] = r.get("LDS_Block_Size")
            for (k, c), v in acc.items():
                res.setdefault(k, {}).setdefault("counters", {})[c] = sum(v) / len(v)
    tot = {"us": 0.0, "fetch_raw": 0.0, "write": 0.0}
    for k, v in sorted(res.items()):
        c = v.get("counters", {})
        tot["us"] += v.get("trace", {}).get("avg_us", 0.0)
        tot["fetch_raw"] += c.get("FETCH_SIZE", 0.0) * 1024
        tot["write"] += c.get("WRITE_SIZE", 0.0) * 1024
    # gfx950: FETCH_SIZE counts half of a streaming read (profiles/README.md)
    res["total"] = {"kernel_us_sum": tot["us"], "fetch_bytes_raw": tot["fetch_raw"], "write_bytes": tot["write"],
                    "hbm_bytes_per_batch": 2 * tot["fetch_raw"] + tot["write"], "frames": 1024,
                    "hbm_bytes_per_frame": (2 * tot["fetch_raw"] + tot["write"]) / 1024, "algorithmic_bytes_per_frame": 160 * 160 * 3 + 20 * 20 * 18}
    try:    # stamp: the id of the library the profile was taken with (bench.py reports these counters only for the same build)
        import importlib
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        res["total"]["source_hash"] = importlib.import_module("stm32h7-yolo_amd").load().yf_network_build_id().decode()
    except Exception as e:      # noqa: BLE001
        res["total"]["source_hash"] = None
        print("no build id:", e)
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
