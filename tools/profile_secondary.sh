#!/bin/bash
# rocprofv3 passes over ONE side configuration of the bench line, taken on the command that produces that line:
#   python3 bench.py --only-secondary <section>     (same clock settle, same timed region as the `secondary` entry of the full line)
# Kernel trace first (default settle and steps), then the counter groups in SEPARATE passes (short: per-launch counts do not depend on the
# clock state; gpurun refuses --pmc combined with trace domains other than --kernel-trace).  Results -> gpurun_out/prof_sec/<section>/<tag>/
#   usage (through gpurun): bash tools/profile_secondary.sh <fp16_56x56|int8_160x160|camera_rgb565_112x112> <tag>
set -u
SEC=${1:?section}; TAG=${2:-run}
OUT=$PWD/gpurun_out/prof_sec/$SEC/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
export YF_NO_BUILD=1     # binding.load() starts no child process under the profiler (no make / sh / sha256sum instrumented by the tool); a library that is not current is refused
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py --only-secondary $SEC > $OUT/bench_line.json 2> $OUT/trace.err
echo "trace rc=$?"
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -o p -- python3 bench.py --only-secondary $SEC --clock-settle-ms 0 --secondary-iters 6 > /dev/null 2> $OUT/pmc$i.err
  echo "pass $i rc=$?"
done
python3 tools/summarize_secondary.py $OUT $SEC > $OUT/summary_print.txt
# the line of an unprofiled run of the same command, for the record next to the traced one
python3 bench.py --only-secondary $SEC > $OUT/bench_line_plain.json 2>> $OUT/trace.err
rm -rf $OUT/trace/*/*.db 2>/dev/null
echo "== $SEC done"
