#!/bin/bash
# The fp16 configuration under rocprofv3: since round 5 a thin front for tools/profile_secondary.sh (trace and counter passes -- SQ_LDS_IDX_ACTIVE,
# SQ_LDS_BANK_CONFLICT, SQ_WAIT_ANY included -- on `python3 bench.py --only-secondary fp16_56x56`, the command that prints the bench line's entry, instead of on
# tools/fp16_bench.py).      usage: tools/profile_fp16.sh <tag>
exec bash "$(dirname "$0")/profile_secondary.sh" fp16_56x56 "${1:-run}"
