#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "lz or LZ or config3" > gpurun_out/aa_tests.txt 2>&1; echo tests rc=$?; tail -2 gpurun_out/aa_tests.txt
timeout 600 python tools/gpu/lz_rounds.py > gpurun_out/aa_lz_rounds.txt 2>&1; echo rc=$?
grep "parse_kernel\|rounds" gpurun_out/aa_lz_rounds.txt | tail -8
timeout 600 python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/aa_bench_lz.json 2> gpurun_out/aa_bench_lz.err; echo rc=$?
python -c "
import json; d=json.loads(open('gpurun_out/aa_bench_lz.json').read().strip().splitlines()[-1]); print(d['value'], d['encode_MBps'], d['decode_MBps'], d.get('roundtrip_ok'))"
