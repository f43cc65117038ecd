#!/bin/bash
# round 3, final build: GPU tests, PMC profile of the headline bench, per-stage counters and stage timeline of the int8 kernel, profiles of
# the fp16 kernel (+ its stage timeline) and of the 160x160 path, the bench line at the driver's flags.  Needs lib_prof (-DYF_BARPROF) and
# lib_f16prof (-DYF16_BARPROF) and lib_f16stage (-DYF16_STAGEPMC) next to lib.     usage (through gpurun): bash tools/r03_profile_final.sh <tag>
set -u
TAG=${1:-r03_e}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/gputest.log
bash tools/profile_pmc.sh $TAG > $O/profile_pmc.log 2>&1; echo "profile_pmc rc=$?"
rm -rf $O/stage_pmc
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA --output-format csv -d $O/stage_pmc -o p -- python3 tools/stage_pmc.py run > /dev/null 2> $O/stage_pmc.err
python3 tools/stage_pmc.py report $O/stage_pmc > $O/stage_pmc.txt 2>&1; tail -3 $O/stage_pmc.txt
YF_LIB_PATH=$PWD/stm32h7-yolo_amd/lib_prof/libyf_network.so python3 tools/barrier_profile.py 2>&1 | grep -v amdgpu.ids > $O/stage_timeline.txt; head -3 $O/stage_timeline.txt
bash tools/profile_fp16.sh $TAG > $O/profile_fp16.log 2>&1; echo "profile_fp16 rc=$?"
YF_LIB_PATH=$PWD/stm32h7-yolo_amd/lib_f16prof/libyf_network.so python3 tools/fp16_timeline.py 2>&1 | grep -v amdgpu.ids > $O/fp16_timeline.txt; head -3 $O/fp16_timeline.txt
bash tools/fp16_stage_pmc.sh $O/fp16_stage_pmc.txt; tail -1 $O/fp16_stage_pmc.txt | cut -c1-200
bash tools/profile_160.sh $TAG > $O/profile_160.log 2>&1; echo "profile_160 rc=$?"
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench.err; echo "bench rc=$?"; cut -c1-400 $O/bench_driver_flags.json
