#!/bin/bash
# round 3, call D: whole GPU suite after the kernel changes of this round, then configs[3] (row stores of the inverse RANK, rANS-1 header parse by
# context, decoder loop), configs[2] (literal-extension chain on the scalar unit, smaller LDS footprint of the segment parse, rounds per block)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/d_pytest.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/d_pytest.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/d_bench_bwt.json 2> gpurun_out/d_bench_bwt.err; echo "bwt rc=$?"
KNZ_LZS_PROF=1 timeout 600 python bench.py --config lz --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/d_bench_lz.json 2> gpurun_out/d_bench_lz.err; echo "lz rc=$?"
python - <<'PY'
import json
for n in ['bwt','lz']:
    try:
        d=json.loads(open(f'gpurun_out/d_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, d['value'], d['encode_MBps'], d['decode_MBps'], d.get('bit_exact_vs_oracle'), d['roofline']['kernel_ms_per_step'], d['roofline']['all_stage_ms'], d.get('fallback_counters_last_batch'))
    except Exception as e: print(n,'ERR',e)
PY
grep "rounds per block" gpurun_out/d_bench_lz.err | tail -1
