#!/bin/bash
# round 3: tests + A/B against the round-2 stage forms + per-stage counters + stage timeline of the lean kernel
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r03a
python -m pytest tests -x -q -m gpu > gpurun_out/r03a/gputest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03a/gputest.log
bash tools/ab_series.sh gpurun_out/r03a/ab.txt lib_v1x | grep -v amdgpu.ids
rm -rf gpurun_out/r03a/stage_pmc
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA"; do
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/r03a/stage_pmc -o p -- python3 tools/stage_pmc.py run > /dev/null 2> gpurun_out/r03a/stage_pmc.err
done
python3 tools/stage_pmc.py report gpurun_out/r03a/stage_pmc > gpurun_out/r03a/stage_pmc.txt 2>&1; cat gpurun_out/r03a/stage_pmc.txt
YF_LIB_PATH=$PWD/stm32h7-yolo_amd/lib_prof/libyf_network.so python3 tools/barrier_profile.py > gpurun_out/r03a/stage_timeline.txt 2>&1; cat gpurun_out/r03a/stage_timeline.txt | grep -v amdgpu.ids
python3 bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err; cat gpurun_out/r03a/bench.json | cut -c1-600
