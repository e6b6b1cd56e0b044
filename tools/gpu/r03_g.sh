#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "LZ or lz or fuzz or rank or RANK" > gpurun_out/g_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/g_pytest.log
timeout 600 python tools/gpu/lz_rounds.py > gpurun_out/g_lz_rounds.txt 2>&1; echo rc=$?; grep "parse_kernel\|rounds" gpurun_out/g_lz_rounds.txt | tail -12
timeout 600 python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/g_bench_lz.json 2> gpurun_out/g_bench_lz.err; echo "lz rc=$?"
timeout 600 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > gpurun_out/g_bench_bwt.json 2> gpurun_out/g_bench_bwt.err; echo "bwt rc=$?"
python - <<'PY'
import json
for n in ['lz','bwt']:
    d=json.loads(open(f'gpurun_out/g_bench_{n}.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['encode_MBps'], d['decode_MBps'], d.get('bit_exact_vs_oracle'), d['roofline']['kernel_ms_per_step'], d['roofline']['all_stage_ms'])
PY
