#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "text_ or utf" --durations=5 > gpurun_out/t_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/t_pytest.log
timeout 900 python bench.py --config l5 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/t_bench_l5.json 2> gpurun_out/t_bench_l5.err; echo "bench rc=$?"; cat gpurun_out/t_bench_l5.json; tail -3 gpurun_out/t_bench_l5.err
