# Compiler flags and the list of device sources: everything the library's BUILD ID is computed from (Makefile: BUILD_ID; binding.py: expected_build_id).
# -Os: the fused kernel is one ~45 KB straight-line loop body and the instruction cache is shared by two CUs; optimising
#   for size (6.4 k instead of 7.7 k instructions) is 3.6 % faster than -O3 (-O2 2.8 %, -O1 1 %, -Oz not smaller).
# -amdgpu-sched-strategy=iterative-ilp: LLVM's iterative ILP machine scheduler; -1.3 % against the default strategy
#   (max-ilp +1 %, max-memory-clause +1.2 %, iterative-minreg +2 %).
# -disable-lsr: without loop strength reduction the address arithmetic of the job loops stays in the form the
#   stage code states it: -1.1 %.  All measured with tools/lib_compare.py on one box (baseline run twice per round).
HIPFLAGS_BASE = --offload-arch=gfx950 -Os -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -mllvm -amdgpu-sched-strategy=iterative-ilp -mllvm -disable-lsr
# the fused fp16 kernel: -O3 (measured faster than the -Os / iterative-ilp flags of the int8 engine) without loop strength reduction (+2.3 %; with
# iterative-ilp or max-ilp on top -1 %)
FP16FLAGS_BASE = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -mllvm -disable-lsr
DEVICE_SRCS = yf_engine.hip yf_kernels.hip.h yf_fused56.hip.h yf_band160.hip.h yf_lab_stages.hip.h yf_lab_layerwise.hip.h yf_decode.hip.h yf_tables.h yf_stream_scratch.h gen/yf_decode_tables_gen.h yf_fp16.hip yf_fp16.h
# the C host layer: everything the library's HOST ID is computed from (Makefile: HOST_ID; binding.py: expected_host_id) -- a library with stale host code must not pass for current on a box without make
HOST_SRCS = network_abi.c platform_abi.c yf_host_prep.c yf_host_prep.h yf_impl.h yf_engine.h yf_fp16.h st_graph_view.h yf_tables.h gen/yf_model_gen.h gen/yf_weights_blob_gen.c ../../include/yf_network.h
