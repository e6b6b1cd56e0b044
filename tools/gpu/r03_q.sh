#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "rank or full_size or bwt" > gpurun_out/q_tests.txt 2>&1; echo tests rc=$?; tail -3 gpurun_out/q_tests.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-pmc > gpurun_out/q_bench.json 2> gpurun_out/q_bench.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/q_bench.json').read().strip().splitlines()[-1])
print(d['value'], d.get('encode_MBps'), d.get('decode_MBps'))
print(d['roofline'].get('kernel_ms_per_step'))
PY
