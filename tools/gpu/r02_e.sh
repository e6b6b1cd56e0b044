#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "full_size_config4 or config4 or entropy_objects or ans1 or corrupt or (stream_bit_exact) or reference_test" > gpurun_out/e_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/e_pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/e_bench.json 2> gpurun_out/e_bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/e_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['encode_MBps'], d['decode_MBps'], d['bit_exact_vs_oracle']); print(d['roofline']['kernel_ms_per_step']); print(d['roofline']['all_stage_ms'])"; tail -2 gpurun_out/e_bench.err
