#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== nproc $(nproc)  $(rocminfo | grep -m1 gfx9)" > gpurun_out/env.log
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/pytest_gpu.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5) > gpurun_out/smoke.log
(timeout 900 python bench.py --steps 5 --warmup 2 2> gpurun_out/bench.err | tail -1) > gpurun_out/bench.json
(timeout 1500 python bench.py --config bwt --steps 3 --warmup 1 2> gpurun_out/bench_bwt.err | tail -1) > gpurun_out/bench_bwt.json
cat gpurun_out/bench.json gpurun_out/bench_bwt.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_bwt -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config bwt --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_bwt.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err)
tail -6 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log; tail -3 gpurun_out/bench_bwt.err
