#!/bin/bash
# VGPR / spill statistics of the fused int8 kernels (every compiled shape).  DEV TOOL, container only.
#   usage: tools/kernel_regs.sh [extra hipcc flags, e.g. -DYF_LAUNDER_X=3]
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -Os -std=c++17 -ffp-contract=off -fPIC -S --cuda-device-only -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -disable-lsr "$@" \
  -I/root/repo/stm32h7-yolo_amd/csrc /root/repo/stm32h7-yolo_amd/csrc/yf_engine.hip -o /tmp/yf_engine.s 2>/dev/null
python3 - <<'PY'
import re
txt = open('/tmp/yf_engine.s').read()
for blk in re.findall(r'- \.agpr_count:.*?\.wavefront_size:\s+\d+', txt, flags=re.S):
    if 'yoloface56_fused' in blk or 'band_k' in blk:
        g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
        print(g('name')[:64], 'vgpr', g('vgpr_count'), 'vgpr_spill', g('vgpr_spill_count'), 'sgpr_spill', g('sgpr_spill_count'), 'scratch', g('private_segment_fixed_size'))
PY
