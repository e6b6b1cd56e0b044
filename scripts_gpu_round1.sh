#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "ANS1 or ans1 or config4 or stream" > gpurun_out/pytest_gpu.log 2>&1
tail -2 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --config bwt --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_bwt_q.json 2> gpurun_out/bench_bwt_q.err
python - <<PY
import json
d=json.loads(open('gpurun_out/bench_bwt_q.json').read().strip().splitlines()[-1])
print(d['value'], d['encode_MBps'], d['decode_MBps'], d['roofline']['all_stage_ms'], d['bit_exact_vs_oracle'])
PY
