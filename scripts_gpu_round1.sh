#!/bin/bash
# one gpurun call: parity tests, smoke, bench(es), rocprof kernel trace
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== nproc $(nproc)  $(rocminfo | grep -m1 gfx9)" > gpurun_out/env.log
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/pytest_gpu.log
(timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5) > gpurun_out/smoke.log
(timeout 900 python bench.py --steps 5 --warmup 2 2> gpurun_out/bench.err | tail -1) > gpurun_out/bench.json
(timeout 900 python bench.py --config ans0 --steps 5 --warmup 2 2> gpurun_out/bench_ans0.err | tail -1) > gpurun_out/bench_ans0.json
cat gpurun_out/bench.json gpurun_out/bench_ans0.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_ans0 -o r1 -- python $GRAFT_REPO_ROOT/bench.py --config ans0 --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_ans0.json 2>> $GRAFT_REPO_ROOT/gpurun_out/prof.err)
tail -4 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log
