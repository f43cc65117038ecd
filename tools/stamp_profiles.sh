#!/bin/bash
# SUPERSEDED (round 4's copy step for tools/r04_profile.sh); round 5 copies by hand from gpurun_out/r05p and gpurun_out/prof_sec (profiles/README.md).
# Copies the results of tools/r04_profile.sh (gpurun_out/r04/, merged back by gpurun) into profiles/ under a tag: the headline files as profiles/r04_<tag>_*,
# the fp16 / 160x160 summaries into their directories, pmc_current.json (what bench.py reads for roofline.traffic; stamped with the build id).  The bench
# lines of that run carry no counter traffic yet (their build's stamp was not in profiles/ when they ran): re-run `python3 bench.py` twice afterwards
# (default flags, --steps 20 --warmup 5) and pass the two lines as $2 / $3 to store them.  DEV TOOL, container side.
#   usage: bash tools/stamp_profiles.sh <tag> [bench_line.json bench_line_driver_flags.json]
set -eu
T=$1; R=gpurun_out/r04; P=profiles
cp $R/int8_bench_kernel_stats.csv $P/r04_${T}_bench_kernel_stats.csv
cp $R/int8_bench_line_profile_run.json $P/r04_${T}_bench_line_profile_run.json
cp $R/int8_pmc_summary.json $P/r04_${T}_pmc_summary.json
cp $R/int8_stage_timeline.txt $P/r04_${T}_stage_timeline.txt
cp $R/pmc_current.json $P/pmc_current.json
cp $R/fp16_summary.json $P/r04_fp16/summary.json; cp $R/fp16_kernel_stats.csv $P/r04_fp16/kernel_stats.csv
cp $R/fp16_stage_pmc.txt $P/r04_fp16/stage_pmc.txt; cp $R/fp16_stage_timeline.txt $P/r04_fp16/stage_timeline.txt; cp $R/fp16_pmc.txt $P/r04_fp16/pmc_counters.txt
cp $R/160_summary.json $P/r04_160/summary.json; cp $R/160_kernel_stats.csv $P/r04_160/kernel_stats.csv
if [ $# -ge 3 ]; then cp $2 $P/r04_${T}_bench_line.json; cp $3 $P/r04_${T}_bench_line_driver_flags.json; fi
python3 - <<PY
import json
d = json.load(open("$P/pmc_current.json")); print("stamped:", d["source_hash"], "trace", round(d["kernel_trace_avg_ns"] / 1e3, 1), "us, events", d["bench_kernel_ms"], "ms, traffic", d["hbm_bytes_per_launch"])
PY
