#!/bin/bash
# Round-6 evidence run (the round-5 script, re-tagged) on the GPU box (through gpurun): every number of the bench line from the command that prints it.
#   int8      headline: rocprofv3 kernel trace + counter passes over bench.py (tools/profile_pmc.sh; two launch streams by default: the summary separates
#             launches that overlap a neighbour from those that do not), and the same trace with --streams 1
#   fp16, 160, camera the side configurations on `bench.py --only-secondary ...` (tools/profile_secondary.sh)
#   bench     the plain lines: default flags, the driver's flags, --streams 1
# Results -> gpurun_out/r06p/     usage: bash tools/r06_profile.sh [part ...]
set -u
export TMPDIR=/tmp
export YF_NO_BUILD=1     # no child processes under the profiler (binding.load)
PARTS=${*:-"int8 fp16 160 camera bench"}
R=gpurun_out/r06p; mkdir -p $R
for P in $PARTS; do
  echo "== $P $(date +%T)"
  case $P in
    int8)  bash tools/profile_pmc.sh r06 > $R/int8_profile.log 2>&1
           cp gpurun_out/prof/r06/summary.json $R/int8_pmc_summary.json; cp gpurun_out/prof/r06/pmc_current.json $R/pmc_current.json
           find gpurun_out/prof/r06/trace -name "*kernel_stats.csv" -exec cp {} $R/int8_bench_kernel_stats.csv \; ; cp gpurun_out/prof/r06/bench_line.json $R/int8_bench_line_profile_run.json
           rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace_s1 -o t -- python3 bench.py --no-secondary --streams 1 > $R/int8_bench_line_profile_run_one_stream.json 2> $R/trace_s1.err
           find $R/trace_s1 -name "*kernel_stats.csv" -exec cp {} $R/int8_bench_kernel_stats_one_stream.csv \; ; rm -rf $R/trace_s1 ;;
    fp16)  bash tools/profile_secondary.sh fp16_56x56 r06 > $R/fp16_profile.log 2>&1 ;;
    160)   bash tools/profile_secondary.sh int8_160x160 r06 > $R/160_profile.log 2>&1 ;;
    camera) bash tools/profile_secondary.sh camera_rgb565_112x112 r06 > $R/camera_profile.log 2>&1 ;;
    bench) python3 bench.py > $R/bench_line.json 2> $R/bench.err; python3 bench.py --steps 20 --warmup 5 > $R/bench_line_driver_flags.json 2>> $R/bench.err
           python3 bench.py --steps 20 --warmup 5 --streams 1 --no-secondary > $R/bench_line_driver_flags_one_stream.json 2>> $R/bench.err ;;
  esac
done
echo "== done $(date +%T)"; ls -la $R
