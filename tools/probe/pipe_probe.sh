#!/bin/bash
# ai_network_run on host arrays: rate per batch size for several (pipeline threshold, chunk, last chunk) settings of yf_engine_run_host (YF_PIPE_MIN /
# YF_PIPE_CHUNK / YF_PIPE_LAST).  YF_PIPE_MIN is read only by a build that turns PIPE_MIN_N of yf_engine.hip into an environment knob (round 5's probe build,
# not in the tree: the shipped 2048 won, profiles/r05_host_path_pipe_probe.txt); on the product only the two other knobs act.  DEV TOOL.   usage (through gpurun): bash tools/probe/pipe_probe.sh [libdir]
L=$PWD/stm32h7-yolo_amd/${1:-lib}/libyf_network.so
for cfg in "2048 3072 1024" "512 1536 512" "512 1536 768" "512 2048 512" "512 2048 1024" "1024 1536 512" "512 1280 512"; do set -- $cfg
  echo "== YF_PIPE_MIN=$1 YF_PIPE_CHUNK=$2 YF_PIPE_LAST=$3"
  YF_LIB_PATH=$L YF_PIPE_MIN=$1 YF_PIPE_CHUNK=$2 YF_PIPE_LAST=$3 YF_NS=512,1024,1536,2048,3072,4096,6144,8192,16384,65535 python tools/probe/host_path_probe.py 2>&1 | grep -v amdgpu
done
