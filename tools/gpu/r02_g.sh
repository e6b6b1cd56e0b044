#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "ans1 or ANS1 or lz or LZ or config4" > gpurun_out/g_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/g_pytest.log
for c in lz bwt; do
timeout 900 python bench.py --config $c --steps 3 --warmup 1 --no-pmc > gpurun_out/g_bench_$c.json 2> gpurun_out/g_bench_$c.err; echo "bench $c rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/g_bench_$c.json'))
print('$c', d['value'], d['encode_MBps'], d['decode_MBps'], d['bit_exact_vs_oracle'], list(d['roofline']['kernel_ms_per_step'].items())[:3])
PY
done
