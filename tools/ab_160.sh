#!/bin/bash
# A/B of differently compiled libraries on the 160x160 path: per-kernel time of the full-batch launches (rocprofv3 kernel trace).
#   usage: tools/ab_160.sh <lib.so> [<lib.so> ...]        results -> gpurun_out/ab160/
set -u
OUT=$PWD/gpurun_out/ab160
mkdir -p $OUT
export TMPDIR=/tmp
i=0
for lib in "$@"; do
  i=$((i+1))
  export YF_LIB_PATH=$PWD/$lib
  rocprofv3 --kernel-trace --output-format csv -d $OUT/t$i -o t -- python3 tests/dev/parity_160.py 6 > $OUT/run$i.log 2> $OUT/err$i.log
  echo "== $lib: $(tail -1 $OUT/run$i.log)"
  python3 - $OUT/t$i <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        for k in ("band_k1", "band_k2", "band_k3", "band_k4"):
            if k in r["Kernel_Name"]:
                d[k].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Grid_Size_X"]), r["VGPR_Count"], r["LDS_Block_Size"]))
tot = 0.0
for k, v in sorted(d.items()):
    big = [x for x in v if x[0] > 40][1:]
    if not big: continue
    avg = sum(x[0] for x in big) / len(big); tot += avg
    print(f"   {k}: {avg:7.1f} us  grid {big[0][1]} vgpr {big[0][2]} lds {big[0][3]}")
print(f"   sum {tot:.1f} us -> {1024 / tot:.3f} M frames/s")
PY
done
