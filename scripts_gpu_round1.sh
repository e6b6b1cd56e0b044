#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "ANS1 or BWT or config4 or transforms or stream" > gpurun_out/pytest_gpu.log 2>&1
tail -2 gpurun_out/pytest_gpu.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bwt -o r1 -- python $R/bench.py --config bwt --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_bwt.json 2> $R/gpurun_out/prof_bwt.err
python - <<PY
import json
d=json.loads(open('$R/gpurun_out/prof_bench_bwt.json').read().strip().splitlines()[-1])
print(d['value'], d['encode_MBps'], d['decode_MBps'], d['roofline']['all_stage_ms'], d['bit_exact_vs_oracle'])
PY
