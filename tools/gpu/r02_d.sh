#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "full_size_config3 or (transform_objects and (LZ or LZX)) or fuzz or next_row or reference_test or concurrent" > gpurun_out/d_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/d_pytest.log
timeout 600 python bench.py --config lz --no-cpu-baseline --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/d_bench_lz.json 2> gpurun_out/d_bench_lz.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/d_bench_lz.json').read().strip().splitlines()[-1]); print(d['value'], d['encode_MBps'], d['decode_MBps'], d['bit_exact_vs_oracle']); print(d['roofline']['kernel_ms_per_step']); print(d['roofline']['all_stage_ms'])"; tail -2 gpurun_out/d_bench_lz.err
