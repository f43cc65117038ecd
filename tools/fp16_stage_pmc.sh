#!/bin/bash
# per-stage counters of the fused fp16 kernel (two --pmc passes), needs stm32h7-yolo_amd/lib_f16stage built with EXTRA_FP16FLAGS=-DYF16_STAGEPMC
#   usage (through gpurun): bash tools/fp16_stage_pmc.sh <out file>
export TMPDIR=/tmp YF_LIB_PATH=$PWD/stm32h7-yolo_amd/lib_f16stage/libyf_network.so
rm -rf gpurun_out/f16stage1 gpurun_out/f16stage2
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA --output-format csv -d gpurun_out/f16stage1 -o p -- python3 tools/fp16_stage_pmc.py run > /dev/null 2> gpurun_out/f16stage1.err &&
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/f16stage2 -o p -- python3 tools/fp16_stage_pmc.py run > /dev/null 2> gpurun_out/f16stage2.err
python3 tools/fp16_stage_pmc.py report gpurun_out/f16stage1 gpurun_out/f16stage2 > $1 2>&1
