#!/bin/bash
# round 3, call C: LZ (literal-extension chain, per-round parse times), rANS order-1 decoder loop A/B
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "LZ or lz or ans1 or ANS1 or config4 or foreign" --durations=4 > gpurun_out/c_pytest.log 2>&1; echo "pytest rc=$?"; tail -7 gpurun_out/c_pytest.log
timeout 600 python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/c_bench_lz.json 2> gpurun_out/c_bench_lz.err; echo "lz rc=$?"
timeout 600 python bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/c_bench_bwt.json 2> gpurun_out/c_bench_bwt.err; echo "bwt rc=$?"
KNZ_ANS1_LDS1=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > gpurun_out/c_bench_bwt_lds1.json 2> gpurun_out/c_bench_bwt_lds1.err; echo "bwt lds1 rc=$?"
python - <<'PY'
import json
for n in ['lz','bwt','bwt_lds1']:
    try:
        d=json.loads(open(f'gpurun_out/c_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, d['value'], d['encode_MBps'], d['decode_MBps'], d.get('bit_exact_vs_oracle'), d['roofline']['kernel_ms_per_step'], d['roofline'].get('kernel_launches_per_step'), d['roofline']['all_stage_ms'], d.get('fallback_counters_last_batch'))
    except Exception as e: print(n,'ERR',e)
PY
