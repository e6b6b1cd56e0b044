#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for w in 0 1024 4096; do echo "WARM=$w"; KNZ_LZS_PROF=1 KNZ_LZ_WARM=$w timeout 600 python tools/gpu/lz_rounds.py 2>&1 | grep "parse_kernel\|rounds" | tail -9 | tr '\n' ' ' | sed 's/knz_lzs_parse_kernel *//g'; echo; done
