#!/bin/bash
# rocprofv3 passes over the fp16 configuration (tools/fp16_bench.py) on the GPU box.  Results -> gpurun_out/prof_fp16/<tag>/
set -u
TAG=${1:-run}
OUT=$PWD/gpurun_out/prof_fp16/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/fp16_bench.py > $OUT/run.log 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -o p -- python3 tools/fp16_bench.py > /dev/null 2> $OUT/pmc$i.err
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
res = {"counters": {}}
for f in glob.glob(os.path.join(out, "trace", "*kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if "f16_fused" in r["Name"]:
            res["kernel"] = r["Name"]; res["trace"] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "min_us": float(r["MinNs"]) / 1e3}
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    acc, disp = {}, {}
    for r in csv.DictReader(open(f)):
        if "f16_fused" in r.get("Kernel_Name", ""):
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            disp.setdefault(r["Counter_Name"], set()).add(r["Dispatch_Id"])
    for k, v in acc.items():
        # one kernel dispatch = several rows (one per XCD/SE): per launch = total / distinct dispatches seen in THIS file
        res["counters"][k] = sum(v) / len(disp[k])
        res.setdefault("launches", {})[k] = len(disp[k])
c = res["counters"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    res["hbm_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024      # gfx950: FETCH_SIZE counts half of a streaming read
    res["algorithmic_bytes_per_launch"] = 4096 * (56 * 56 * 3 * 2 + 7 * 7 * 18 * 4)
try:    # stamp: the id of the library the profile was taken with (bench.py reports these counters only for the same build)
    import importlib
    sys.path.insert(0, os.getcwd())
    res["source_hash"] = importlib.import_module("stm32h7-yolo_amd").load().yf_network_build_id().decode()
except Exception as e:
    res["source_hash"] = None
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
