#!/bin/bash
# rocprofv3 passes over the 160x160 path (tests/dev/parity_160.py) on the GPU box: kernel trace, then HBM counters in
# separate passes.  Results -> gpurun_out/prof160/<tag>/      usage: tools/profile_160.sh <tag>
set -u
TAG=${1:-run}
OUT=$PWD/gpurun_out/prof160/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tests/dev/parity_160.py 6 > $OUT/run.log 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -o p -- python3 tests/dev/parity_160.py 6 > /dev/null 2> $OUT/pmc$i.err
  echo "pass $i rc=$?"
done
python3 tools/summarize_pmc160.py $OUT
