#!/bin/bash
# interleaved A/B (tools/ab_bench.py) of the shipped kernel against the experimental build of several library variants
#   usage: tools/ab_series.sh <out file> <lib dir> [<lib dir> ...]     (directories under stm32h7-yolo_amd/)
out=$1; shift
: > $out
for d in "$@"; do
  echo "== $d" >> $out
  YF_LIB_PATH=$PWD/stm32h7-yolo_amd/$d/libyf_network.so timeout -k 10 180 python3 tools/ab_bench.py >> $out 2>&1 || echo "FAILED rc=$?" >> $out
done
cat $out
