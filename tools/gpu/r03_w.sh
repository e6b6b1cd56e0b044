#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "rank or full_size" > gpurun_out/w_tests.txt 2>&1; echo tests rc=$?; tail -2 gpurun_out/w_tests.txt
KNZ_RANK_PROF=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/w_bench.json 2> gpurun_out/w_bench.err; echo rc=$?
grep "inverse RANK chain" gpurun_out/w_bench.err | tail -1
python -c "
import json; d=json.loads(open('gpurun_out/w_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['encode_MBps'], d['decode_MBps'], d.get('roundtrip_ok'))"
