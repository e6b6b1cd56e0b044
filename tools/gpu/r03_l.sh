#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
KNZ_LZS_PROF=1 timeout 600 python tools/gpu/lz_rounds.py > gpurun_out/l_lz_rounds.txt 2>&1; echo rc=$?
grep "LZ forward round" gpurun_out/l_lz_rounds.txt | tail -5 | cut -c1-1500
