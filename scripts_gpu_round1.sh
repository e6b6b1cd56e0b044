#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/pytest_gpu.log
(timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench.err | tail -1) > gpurun_out/bench.json
cat gpurun_out/bench.json
(timeout 900 python bench.py --config ans0 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench_ans0.err | tail -1) > gpurun_out/bench_ans0.json
cat gpurun_out/bench_ans0.json
tail -3 gpurun_out/pytest_gpu.log
