#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "lz" > gpurun_out/m_tests.txt 2>&1; echo tests rc=$?; tail -3 gpurun_out/m_tests.txt
KNZ_LZS_PROF=1 timeout 600 python tools/gpu/lz_rounds.py > gpurun_out/m_lz_rounds.txt 2>&1; echo rc=$?
grep "LZ forward round" gpurun_out/m_lz_rounds.txt | tail -6 | cut -c1-900
grep "ms$\|rounds" gpurun_out/m_lz_rounds.txt | tail -40
KNZ_LZS_ALL_AGAIN=1 timeout 600 python tools/gpu/lz_rounds.py > gpurun_out/m_lz_rounds_all.txt 2>&1; echo rc=$?
grep "parse\|rounds" gpurun_out/m_lz_rounds_all.txt | tail -12
timeout 600 python bench.py --config lz --steps 3 --warmup 1 > gpurun_out/m_bench_lz.json 2> gpurun_out/m_bench_lz.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/m_bench_lz.json').read().strip().splitlines()[-1])
print(d['value'], d.get('encode_MBps'), d.get('decode_MBps'), {k:v for k,v in d.items() if 'MB' in k or 'encode' in k or 'decode' in k})
PY
