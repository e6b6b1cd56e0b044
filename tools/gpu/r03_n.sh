#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
KNZ_LZS_PROF=1 timeout 600 python tools/gpu/lz_rounds.py > gpurun_out/n_lz_rounds.txt 2>&1; echo rc=$?
grep "last parse of each" gpurun_out/n_lz_rounds.txt | tail -1 | cut -c1-2500
grep "parse_kernel" gpurun_out/n_lz_rounds.txt | tail -6
