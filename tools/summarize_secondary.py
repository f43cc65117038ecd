#!/usr/bin/env python3
"""Collapse the rocprofv3 CSVs of tools/profile_secondary.sh (taken on `bench.py --only-secondary <section>`) into summary.json.

   usage: summarize_secondary.py <out_dir> <fp16_56x56|int8_160x160|camera_rgb565_112x112>

   Kernel time comes in two forms, both over FULL-BATCH launches only: `avg_us` over every launch of the command (warm-up and clock-settle
   launches included) and `timed_avg_us` over the LAST `timed_steps` launches -- the ones between the bench's two HIP events.  The line the
   same command printed (bench_line.json) is stored beside them with the check VERDICT round 4 asked for: kernel_us(trace) <= ms_per_step x 1.01.
   Counters are per launch: total over a counter's rows / distinct dispatches (a dispatch may come as several rows)."""
import csv
import glob
import json
import os
import sys

SECTIONS = {"fp16_56x56": (("f16_fused",), 4096, 56 * 56 * 3 * 2 + 7 * 7 * 18 * 4),
            "int8_160x160": (("band_k1", "band_k23", "band_k4"), 1024, 160 * 160 * 3 + 20 * 20 * 18),
            # the camera-input build of the fused kernel (template arguments <2, 8, false, true>); the section also times the two-launch form, whose launches
            # (prepare_rgb565_kernel + the int8-input build) do not carry this name
            "camera_rgb565_112x112": (("8, false, true>",), 4096, 112 * 112 * 2 + 7 * 7 * 18)}


def short(name, keys):
    for k in sorted(keys, key=len, reverse=True):       # band_k23 before band_k2
        if k in name:
            return k
    return None


def main():
    out, section = sys.argv[1], sys.argv[2]
    keys, frames, algo = SECTIONS[section]
    line = json.loads(open(os.path.join(out, "bench_line.json")).read().strip().splitlines()[-1])
    sec = line["secondary"][section]
    timed = int(sec["timed_steps"])
    res = {"section": section, "command": f"python3 bench.py --only-secondary {section}", "bench_line_of_the_traced_run": sec, "kernels": {}}
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"], keys)
            if k:
                per.setdefault(k, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Grid_Size_X"]), r["Kernel_Name"]))
        for k, v in per.items():
            full = max(g for _, _, g, _ in v)
            d = [dur / 1e3 for _, dur, g, _ in sorted(v) if g == full]
            t = d[-timed:]
            res["kernels"][k] = {"name": v[0][3], "trace": {"calls": len(d), "avg_us": sum(d) / len(d), "min_us": min(d),
                                                            "timed_calls": len(t), "timed_avg_us": sum(t) / len(t), "timed_min_us": min(t), "timed_max_us": max(t)}}
    for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = [(short(r.get("Kernel_Name", ""), keys), r) for r in csv.DictReader(open(f))]
            rows = [(k, r) for k, r in rows if k]
            full = {}
            for k, r in rows:
                full[k] = max(full.get(k, 0), int(r["Grid_Size"]))
            acc, disp = {}, {}
            for k, r in rows:
                if int(r["Grid_Size"]) != full[k]:
                    continue
                c = r["Counter_Name"]
                acc[(k, c)] = acc.get((k, c), 0.0) + float(r["Counter_Value"])
                disp.setdefault((k, c), set()).add(r["Dispatch_Id"])
                kk = res["kernels"].setdefault(k, {})
                for key in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Grid_Size", "Workgroup_Size"):
                    if key in r:
                        kk.setdefault("dispatch", {})[key] = r[key]
            for (k, c), v in acc.items():
                res["kernels"][k].setdefault("counters", {})[c] = v / len(disp[(k, c)])
                res["kernels"][k].setdefault("counter_launches", {})[c] = len(disp[(k, c)])
    # derived, per kernel: the pipes VERDICT asked about (LDS pipe busy, bank-conflict share, wait share, VALU issue share)
    tot = {"us": 0.0, "timed_us": 0.0, "fetch_raw": 0.0, "write": 0.0}
    for k, v in res["kernels"].items():
        c, tr = v.get("counters", {}), v.get("trace", {})
        tot["us"] += tr.get("avg_us", 0.0)
        tot["timed_us"] += tr.get("timed_avg_us", 0.0)
        tot["fetch_raw"] += c.get("FETCH_SIZE", 0.0) * 1024
        tot["write"] += c.get("WRITE_SIZE", 0.0) * 1024
        der = {}
        if tr.get("timed_avg_us"):
            cu_cycles = 256 * 2.4e9 * tr["timed_avg_us"] * 1e-6          # one LDS pipe per CU at the nominal 2.4 GHz
            if "SQ_LDS_IDX_ACTIVE" in c:
                der["lds_pipe_busy"] = c["SQ_LDS_IDX_ACTIVE"] / cu_cycles
            if "SQ_INSTS_VALU" in c:
                der["valu_issue_busy_at_4_cycles"] = c["SQ_INSTS_VALU"] * 4.0 / (4 * cu_cycles)
            if "SQ_INSTS_VALU" in c:
                der["valu_instructions_per_frame"] = c["SQ_INSTS_VALU"] / frames
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            der["lds_bank_conflict_share"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        if "SQ_WAIT_ANY" in c and c.get("SQ_WAVE_CYCLES"):
            der["wait_any_share_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
        if "SQ_ACTIVE_INST_VALU" in c and c.get("SQ_BUSY_CYCLES"):
            der["note_active_inst_valu_over_busy_cycles"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"]
        if der:
            v["derived"] = der
    hbm = 2 * tot["fetch_raw"] + tot["write"]                             # gfx950: FETCH_SIZE counts half of a streaming read (profiles/README.md)
    step_us = sec["ms_per_step"] * 1e3
    total = {"kernel_us_sum": tot["us"], "timed_kernel_us_sum": tot["timed_us"], "bench_step_us_of_the_traced_run": step_us,
             "kernel_le_step_x_1.01": bool(tot["timed_us"] <= step_us * 1.01),
             "frac_from_trace": frames * algo / (tot["timed_us"] * 1e-6) / 8.0e12 if tot["timed_us"] else None, "frac_of_the_line": sec["roofline"]["frac"],
             "fetch_bytes_raw": tot["fetch_raw"], "write_bytes": tot["write"], "hbm_bytes_per_batch": hbm, "frames": frames,
             "hbm_bytes_per_frame": hbm / frames, "algorithmic_bytes_per_frame": algo, "source_hash": sec.get("kernel_source_hash")}
    res["total"] = total
    if section == "fp16_56x56":        # flat keys bench.py reads for this section
        k = res["kernels"].get("f16_fused", {})
        res["kernel"], res["trace"], res["counters"] = k.get("name"), k.get("trace"), k.get("counters", {})
        res["hbm_bytes_per_launch"], res["algorithmic_bytes_per_launch"], res["source_hash"] = hbm, frames * algo, total["source_hash"]
    json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
    print(json.dumps({"total": total, "kernels": {k: {"trace": v.get("trace"), "derived": v.get("derived")} for k, v in res["kernels"].items()}}, indent=1))


if __name__ == "__main__":
    main()
