#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "rank" > gpurun_out/v_tests.txt 2>&1; echo tests rc=$?; tail -2 gpurun_out/v_tests.txt
for v in 9; do
  KNZ_RANK_VARIANT=$v KNZ_RANK_PROF=1 timeout 900 python bench.py --steps 1 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/v_bench_$v.json 2> gpurun_out/v_bench_$v.err; echo variant $v rc=$?
  grep "inverse RANK chain" gpurun_out/v_bench_$v.err | tail -1
  python -c "
import json; d=json.loads(open('gpurun_out/v_bench_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['decode_MBps'], d.get('roundtrip_ok'))"
done
