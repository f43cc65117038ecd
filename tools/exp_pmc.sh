#!/bin/bash
# HBM-side counters of the shipped and the experimental build of one library:  tools/exp_pmc.sh <lib dir under stm32h7-yolo_amd/> [counters...]
set -u
export TMPDIR=/tmp
D=$1; shift
CTRS=${*:-"FETCH_SIZE WRITE_SIZE"}
OUT=gpurun_out/exp_pmc/$D; mkdir -p $OUT
for c in $CTRS; do
  for e in 0 1; do
    YF_EXPERIMENTAL=$e YF_LIB_PATH=$PWD/stm32h7-yolo_amd/$D/libyf_network.so rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${c}_$e -o p -- python3 tools/exp_run.py > /dev/null 2> $OUT/${c}_$e.err || echo "pass $c $e failed"
  done
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
for d in sorted(glob.glob(sys.argv[1] + "/*_[01]")):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "yoloface56_fused" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(d.split("/")[-1], k, "per launch %.0f" % (sum(v) / len(v)), "launches", len(v))
PY
