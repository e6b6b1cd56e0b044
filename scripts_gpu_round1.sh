#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 2000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/pytest_gpu.log
(timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench.err | tail -1) > gpurun_out/bench.json
cat gpurun_out/bench.json
tail -5 gpurun_out/pytest_gpu.log
