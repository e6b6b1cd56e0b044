#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "LZ or lz or fuzz or next_row" --durations=5 > gpurun_out/l_pytest.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/l_pytest.log
timeout 900 python bench.py --config lz --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/l_bench_lz.json 2> gpurun_out/l_bench_lz.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/l_bench_lz.json'))
print(d['value'], d['encode_MBps'], d['decode_MBps'], d['bit_exact_vs_oracle'], d['roofline']['kernel_ms_per_step'], d['roofline']['all_stage_ms'])
PY
tail -3 gpurun_out/l_bench_lz.err
