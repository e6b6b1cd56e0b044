#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_q.json').read().strip().splitlines()[-1])
print(d['value'], d['encode_MBps'], d['decode_MBps'], d['roofline']['all_stage_ms'], d['bit_exact_vs_oracle'], d['roundtrip_ok'])
PY
