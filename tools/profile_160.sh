#!/bin/bash
# The 160x160 path under rocprofv3: since round 5 a thin front for tools/profile_secondary.sh, which takes the kernel trace and the counter passes (FETCH_SIZE,
# WRITE_SIZE, the SQ instruction counters, SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT / SQ_WAIT_ANY ...) on the command that prints the bench line's
# `secondary.int8_160x160` entry -- `python3 bench.py --only-secondary int8_160x160` -- instead of on tests/dev/parity_160.py, whose six launches ran inside
# the GPU's clock ramp (rounds 2-4: the trace's kernel times exceeded the bench step).      usage: tools/profile_160.sh <tag>
exec bash "$(dirname "$0")/profile_secondary.sh" int8_160x160 "${1:-run}"
