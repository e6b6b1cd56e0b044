#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "BWT or bwt or config4 or fuzz or LZ or lz or sort" > gpurun_out/j_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/j_pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/j_bench_bwt.json 2> gpurun_out/j_bench_bwt.err; echo "bwt rc=$?"
KNZ_SORT_THREE_LAUNCH=1 timeout 600 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > gpurun_out/j_bench_bwt3.json 2> gpurun_out/j_bench_bwt3.err; echo "bwt3 rc=$?"
timeout 600 python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/j_bench_lz.json 2> gpurun_out/j_bench_lz.err; echo "lz rc=$?"
python - <<'PY'
import json
for n in ['bwt','bwt3','lz']:
    d=json.loads(open(f'gpurun_out/j_bench_{n}.json').read().strip().splitlines()[-1])
    print(n, d['value'], d['encode_MBps'], d['decode_MBps'], d.get('bit_exact_vs_oracle'), d['roofline']['all_stage_ms'])
PY
