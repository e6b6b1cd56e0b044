#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "BWT or bwt or config4 or fuzz" > gpurun_out/i_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/i_pytest.log
KNZ_BWT_PROF=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/i_bench_bwt.json 2> gpurun_out/i_bench_bwt.err; echo "bwt rc=$?"
grep "suffix sort" gpurun_out/i_bench_bwt.err | head -20
python - <<'PY'
import json
d=json.loads(open('gpurun_out/i_bench_bwt.json').read().strip().splitlines()[-1])
print(d['value'], d['encode_MBps'], d['decode_MBps'], d.get('bit_exact_vs_oracle'), d['roofline']['all_stage_ms'])
PY
