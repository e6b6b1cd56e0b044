#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/y_tests.txt 2>&1; echo tests rc=$?; tail -3 gpurun_out/y_tests.txt
for cfg in l5 lz; do
timeout 600 python bench.py --config $cfg --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/y_bench_$cfg.json 2> gpurun_out/y_bench_$cfg.err; echo $cfg rc=$?
python -c "
import json; d=json.loads(open('gpurun_out/y_bench_$cfg.json').read().strip().splitlines()[-1]); print(d['value'], d['encode_MBps'], d['decode_MBps'], d.get('roundtrip_ok'))"
done
