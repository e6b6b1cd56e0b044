#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for seg in 2048 4096 8192; do
  KNZ_LZ_SEG=$seg timeout 600 python tools/gpu/lz_rounds.py > gpurun_out/o_lz_rounds_$seg.txt 2>&1; echo seg=$seg rc=$?
  grep "rounds\|lzs_" gpurun_out/o_lz_rounds_$seg.txt | tail -30
done
