#!/bin/bash
# SUPERSEDED by tools/r05_profile.sh (round 5: the side configurations are profiled on `bench.py --only-secondary ...`); kept because profiles/README.md names it for the r04 tags.
# Round-4 evidence run on the GPU box (through gpurun): the headline int8 kernel (kernel trace + counter passes over bench.py), its stage timeline, the
# fp16 kernel (trace, counters, per-stage counters, stage timeline), the 160x160 path, then the default bench line.  Results -> gpurun_out/r04/
#   usage: bash tools/r04_profile.sh [part ...]     parts: int8 timeline fp16 f16stage 160 bench   (default: all)
set -u
export TMPDIR=/tmp
PARTS=${*:-"int8 timeline fp16 f16stage 160 bench"}
R=gpurun_out/r04; mkdir -p $R
for P in $PARTS; do
  echo "== $P $(date +%T)"
  case $P in
    int8)     bash tools/profile_pmc.sh r04 > $R/int8_profile.log 2>&1; cp gpurun_out/prof/r04/summary.json $R/int8_pmc_summary.json 2>/dev/null; cp gpurun_out/prof/r04/pmc_current.json $R/pmc_current.json 2>/dev/null
              find gpurun_out/prof/r04/trace -name "*kernel_stats.csv" -exec cp {} $R/int8_bench_kernel_stats.csv \; ; cp gpurun_out/prof/r04/bench_line.json $R/int8_bench_line_profile_run.json ;;
    timeline) YF_LIB_PATH=$PWD/stm32h7-yolo_amd/lib_prof/libyf_network.so python3 tools/barrier_profile.py > $R/int8_stage_timeline.txt 2>&1 ;;
    fp16)     bash tools/profile_fp16.sh r04 > $R/fp16_profile.log 2>&1; cp gpurun_out/prof_fp16/r04/summary.json $R/fp16_summary.json; cp gpurun_out/prof_fp16/r04/kernel_stats.csv $R/fp16_kernel_stats.csv
              YF_LIB_PATH=$PWD/stm32h7-yolo_amd/lib_f16prof/libyf_network.so python3 tools/fp16_timeline.py > $R/fp16_stage_timeline.txt 2>&1
              bash tools/fp16_pmc.sh lib > $R/fp16_pmc.txt 2>&1 ;;
    f16stage) bash tools/fp16_stage_pmc.sh $R/fp16_stage_pmc.txt ;;
    160)      bash tools/profile_160.sh r04 > $R/160_profile.log 2>&1; cp gpurun_out/prof160/r04/summary.json $R/160_summary.json; cp gpurun_out/prof160/r04/kernel_stats.csv $R/160_kernel_stats.csv ;;
    bench)    python3 bench.py > $R/bench_line.json 2> $R/bench.err; python3 bench.py --steps 20 --warmup 5 > $R/bench_line_driver_flags.json 2>> $R/bench.err ;;
  esac
done
echo "== done $(date +%T)"; ls -la $R
