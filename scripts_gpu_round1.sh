#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q > gpurun_out/pytest_gpu.log 2>&1
tail -2 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
timeout 600 python bench.py --config ans0 --steps 3 --warmup 1 > gpurun_out/bench_ans0.json 2> gpurun_out/bench_ans0.err
timeout 900 python bench.py --config lz --steps 1 --warmup 1 > gpurun_out/bench_lz.json 2> gpurun_out/bench_lz.err
for f in bench bench_ans0 bench_lz; do python - <<PY
import json
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['encode_MBps'], d['decode_MBps'], d.get('cpu_baseline',{}).get('encode_MBps'), d.get('cpu_baseline',{}).get('decode_MBps'), d['roofline']['kernel'], d['roofline']['frac'], d['bit_exact_vs_oracle'])
PY
done
