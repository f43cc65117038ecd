set -e
mkdir -p gpurun_out
: > gpurun_out/r03_whatif.txt
for e in 2 3 4 10 11 12; do
  echo "== what-if $e" >> gpurun_out/r03_whatif.txt
  YF_LIB_PATH=$PWD/stm32h7-yolo_amd/lib_exp$e/libyf_network.so timeout -k 10 120 python3 tools/ab_bench.py >> gpurun_out/r03_whatif.txt 2>&1
done
cat gpurun_out/r03_whatif.txt
