#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run through gpurun).  Kernel trace and every PMC group are SEPARATE
# runs (gpurun refuses --pmc combined with trace domains other than --kernel-trace).  Results -> gpurun_out/prof/<tag>/
#   usage: tools/profile_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export YF_NO_BUILD=1     # binding.load() starts no child process under the profiler (no make / sh / sha256sum instrumented by the tool); a library that is not current is refused
TRACE_ARGS="--no-secondary $*"        # kernel trace: the default bench command (400 steps after 100 warm-up), headline kernel only
ARGS="--steps 20 --warmup 5 --no-secondary $*"   # counter passes: per-launch counts do not depend on the clock state
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 bench.py $TRACE_ARGS > $OUT/bench_line.json 2> $OUT/trace.err
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
  "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_WAVES" \
  "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc$i -o p -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc$i.err
  echo "pass $i rc=$?"
done
python3 tools/summarize_pmc.py $OUT
