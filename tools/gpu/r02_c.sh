#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "list_ranking or config4 or full_size_config4 or c_smoke or reference_test or (transform_objects and BWT) or corrupt" > gpurun_out/c_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/c_pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/c_bench.json 2> gpurun_out/c_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/c_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['encode_MBps'], d['decode_MBps'], d['bit_exact_vs_oracle']); print(d['roofline']['kernel_ms_per_step']); print(d['roofline']['all_stage_ms'])"; tail -2 gpurun_out/c_bench.err
