#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "ANS1 or RANK or MTFT or config4 or transforms" > gpurun_out/pytest_gpu.log 2>&1
tail -3 gpurun_out/pytest_gpu.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bwt -o r1 -- python $R/bench.py --config bwt --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_bwt.json 2> $R/gpurun_out/prof_bwt.err
tail -1 $R/gpurun_out/prof_bench_bwt.json
