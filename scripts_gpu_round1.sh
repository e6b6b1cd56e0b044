#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "HUFFMAN or stream or config2 or stress or entropy or decoder_paths" > gpurun_out/pytest_gpu.log 2>&1
tail -2 gpurun_out/pytest_gpu.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
tail -1 $R/gpurun_out/prof_bench.json | cut -c1-300
