#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
KNZ_LZS_PROF=1 timeout 600 python tools/gpu/lz_rounds.py > gpurun_out/f_lz_rounds.txt 2>&1; echo rc=$?; tail -25 gpurun_out/f_lz_rounds.txt
timeout 600 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > gpurun_out/f_bench_bwt.json 2> gpurun_out/f_bench_bwt.err; echo "bwt rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/f_bench_bwt.json').read().strip().splitlines()[-1])
print(d['value'], d['encode_MBps'], d['decode_MBps'], d['roofline']['kernel_ms_per_step'])
PY
