#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for seg in 2048 4096 8192 16384; do
  KNZ_LZ_SEG=$seg timeout 600 python bench.py --config lz --steps 3 --warmup 1 > gpurun_out/p_bench_lz_$seg.json 2> gpurun_out/p_bench_lz_$seg.err; echo seg=$seg rc=$?
  python - $seg <<'PY'
import json,sys
d=json.loads(open('gpurun_out/p_bench_lz_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(d['value'], d.get('encode_MBps'), d.get('decode_MBps'))
PY
done
