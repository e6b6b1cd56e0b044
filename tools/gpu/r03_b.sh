#!/bin/bash
# round 3, call B: segment-parallel LZ forward + the LDS-staged literal-extension chain of the parallel LZ inverse
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_gpu.py -x -q -k "LZ or lz or fuzz or corrupt" --durations=6 > gpurun_out/b_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/b_pytest.log
timeout 900 python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/b_bench_lz.json 2> gpurun_out/b_bench_lz.err; echo "lz rc=$?"
python - <<'PY'
import json
for n in ['lz']:
    try:
        d=json.loads(open(f'gpurun_out/b_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, d['value'], d['encode_MBps'], d['decode_MBps'], d['bit_exact_vs_oracle'], d['roofline']['kernel_ms_per_step'], d['roofline'].get('kernel_launches_per_step'), d['roofline']['all_stage_ms'])
    except Exception as e: print(n,'ERR',e)
PY
tail -n 3 gpurun_out/b_bench_lz.err
