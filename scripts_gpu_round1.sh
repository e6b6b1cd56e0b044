#!/bin/bash
# corrupt-stream test on the GPU (bounded by its own timeout), then the config-4 profile for profiles/ v6
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k corrupt > gpurun_out/pytest_corrupt.log 2>&1; echo "corrupt rc=$?"
tail -5 gpurun_out/pytest_corrupt.log
timeout 600 python bench.py --config bwt --steps 3 --warmup 1 > gpurun_out/bench_bwt_v6.json 2> gpurun_out/bench_bwt_v6.err; echo "bench rc=$?"
cat gpurun_out/bench_bwt_v6.json
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_bwt
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bwt -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config bwt --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/gpurun_out/prof_bwt.log 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bwt.err; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT && python tools/rocpd_summary.py gpurun_out/prof_bwt > gpurun_out/prof_bwt_summary.md 2>&1; head -30 gpurun_out/prof_bwt_summary.md
