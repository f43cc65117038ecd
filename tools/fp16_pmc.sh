#!/bin/bash
# SQ counters of the fused fp16 kernel for several library builds (two rocprofv3 --pmc passes each over tools/fp16_bench.py).  DEV TOOL.
#   usage (through gpurun): bash tools/fp16_pmc.sh <libdir> ...        -> gpurun_out/f16pmc/<libdir>.txt
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/f16pmc
G1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
G2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
for L in "$@"; do
  export YF_LIB_PATH=$PWD/stm32h7-yolo_amd/$L/libyf_network.so
  rm -rf gpurun_out/f16pmc/$L; mkdir -p gpurun_out/f16pmc/$L
  i=0
  for grp in "$G1" "$G2"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/f16pmc/$L/p$i -o p -- python3 tools/fp16_bench.py > gpurun_out/f16pmc/$L/run$i.log 2> gpurun_out/f16pmc/$L/err$i.log
  done
  python3 - $L <<'PY' | tee gpurun_out/f16pmc/$L.txt
import csv, glob, sys, collections
L = sys.argv[1]
acc = collections.defaultdict(float); disp = collections.defaultdict(set)
for f in glob.glob(f"gpurun_out/f16pmc/{L}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "f16_fused" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
print(f"== {L}: per launch of 4096 frames (per frame in brackets)")
for k in sorted(acc):
    v = acc[k] / len(disp[k])
    print(f"   {k:24s} {v/1e6:10.3f} M   [{v/4096:9.1f}]")
PY
done
