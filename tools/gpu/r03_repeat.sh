#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for k in 1 2 3 4; do
  timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "rank or ans1 or full_size or stream_bit_exact" > gpurun_out/rep_$k.txt 2>&1; echo "run $k rc=$?"; tail -1 gpurun_out/rep_$k.txt
done
