#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for mode in default unpacked; do
  if [ $mode = unpacked ]; then export KNZ_RANK_UNPACKED=1; fi
  KNZ_RANK_PROF=1 timeout 900 python bench.py --steps 1 --warmup 1 --no-pmc > gpurun_out/r_bench_$mode.json 2> gpurun_out/r_bench_$mode.err; echo $mode rc=$?
  grep "inverse RANK chain" gpurun_out/r_bench_$mode.err | tail -1
done
