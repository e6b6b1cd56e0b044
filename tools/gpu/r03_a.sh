#!/bin/bash
# round 3, call A: the parallel LZ inverse on the GPU (LZ tests incl. the full-size configs[2] stream, fuzz, damaged streams), its bench line,
# and this round's starting point for configs[3]
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -k "LZ or lz or fuzz or corrupt" --durations=6 > gpurun_out/a_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/a_pytest.log
timeout 900 python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/a_bench_lz.json 2> gpurun_out/a_bench_lz.err; echo "lz rc=$?"
timeout 900 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/a_bench_bwt.json 2> gpurun_out/a_bench_bwt.err; echo "bwt rc=$?"
python - <<'PY'
import json
for n in ['lz','bwt']:
    try:
        d=json.loads(open(f'gpurun_out/a_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, d['value'], d['encode_MBps'], d['decode_MBps'], d['bit_exact_vs_oracle'], d['roofline']['kernel_ms_per_step'], d['roofline']['all_stage_ms'])
    except Exception as e: print(n,'ERR',e)
PY
tail -3 gpurun_out/a_bench_lz.err gpurun_out/a_bench_bwt.err
