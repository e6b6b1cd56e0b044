#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.json 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/gpurun_out/pmc_write.json 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_write.err
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err
tail -2 $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch.err
