#!/bin/bash
# SQ_INSTS_VALU / SQ_INSTS_LDS per launch of the fused int8 kernel (network + fused decode, 4096 frames) for several library builds.  DEV TOOL.
#   usage (through gpurun): bash tools/probe/valu_count.sh <libdir> ...
export TMPDIR=/tmp
for L in "$@"; do
rm -rf gpurun_out/vc
YF_LIB_PATH=$PWD/stm32h7-yolo_amd/$L/libyf_network.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/vc -o p -- python3 tools/probe/decode_cost.py > gpurun_out/vc.log 2>&1
python3 - "$L" <<EOF
import csv,glob,collections,sys
d=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/vc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "yoloface56_fused" in r["Kernel_Name"]: d[r["Counter_Name"]][int(r["Dispatch_Id"])]+=float(r["Counter_Value"])
out=[]
for k,v in sorted(d.items()):
    vals=sorted(v.values()); lo=vals[:len(vals)//2]; hi=vals[len(vals)//2:]
    out.append("%s %.2f M (without decode) / %.2f M (with)" % (k[3:], sum(lo)/len(lo)/1e6, sum(hi)/len(hi)/1e6))
print(sys.argv[1], "; ".join(out))
EOF
done
