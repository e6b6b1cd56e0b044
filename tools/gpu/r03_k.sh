#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "rank or RANK" > gpurun_out/k_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/k_pytest.log
VARIANTS=5,6 timeout 900 python tools/gpu/rank_variants.py > gpurun_out/k_rank_variants.json 2> gpurun_out/k_rank_variants.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/k_rank_variants.json').read().strip().splitlines()[-1])
for b in d['blocks']:
    if len(b['ms'])>1: print(b['block'], b['zero_frac'], b['ge64_frac'], b['ms'])
print('slowest default', d['slowest_block_ms_default'])
PY
KNZ_RANK_VARIANT=6 timeout 600 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/k_bench_bwt6.json 2> gpurun_out/k_bench_bwt6.err; echo "bwt6 rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/k_bench_bwt6.json').read().strip().splitlines()[-1])
print(d['value'], d['encode_MBps'], d['decode_MBps'], d.get('bit_exact_vs_oracle'), d['roofline']['kernel_ms_per_step'])
PY
