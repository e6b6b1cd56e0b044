#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "ans1 or ANS1 or entropy or full_size or corrupt" > gpurun_out/x_tests.txt 2>&1; echo tests rc=$?; tail -2 gpurun_out/x_tests.txt
timeout 900 python bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/x_bench.json 2> gpurun_out/x_bench.err; echo rc=$?
python -c "
import json; d=json.loads(open('gpurun_out/x_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['encode_MBps'], d['decode_MBps'], d.get('roundtrip_ok')); print(d['roofline']['kernel_ms_per_step'])"
