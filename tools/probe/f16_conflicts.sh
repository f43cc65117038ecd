#!/bin/bash
# LDS bank-conflict counters of the fp16 kernel for several library builds (what-if attribution).  DEV TOOL.   usage: f16_conflicts.sh <libdir> ...
export TMPDIR=/tmp
for L in "$@"; do
rm -rf gpurun_out/f16lds
YF_LIB_PATH=$PWD/stm32h7-yolo_amd/$L/libyf_network.so rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --output-format csv -d gpurun_out/f16lds -o p -- python3 tools/fp16_bench.py > gpurun_out/f16lds.log 2>&1
echo "== $L  $(grep frames/s gpurun_out/f16lds.log)"
python3 - <<EOF
import csv,glob,collections
d=collections.defaultdict(list)
for f in glob.glob("gpurun_out/f16lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "f16_fused" in r["Kernel_Name"]: d[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for k,v in sorted(d.items()):
    per=collections.defaultdict(float)
    for i,x in v: per[i]+=x
    vals=sorted(per.values()); print("   %-24s %12.0f" % (k, sum(vals)/len(vals)))
EOF
done
