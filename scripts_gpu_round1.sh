#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q > gpurun_out/pytest_gpu.log 2>&1
tail -2 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err
timeout 900 python bench.py --config bwt --steps 2 --warmup 1 > gpurun_out/bench_bwt.json 2> gpurun_out/bench_bwt.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bwt -o r1 -- python $R/bench.py --config bwt --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/prof_bench_bwt.json 2> $R/gpurun_out/prof_bwt.err
cd $R
for f in bench bench_bwt; do python - <<PY
import json
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['encode_MBps'], d['decode_MBps'], d.get('cpu_baseline',{}).get('encode_MBps'), d.get('cpu_baseline',{}).get('decode_MBps'), d['roofline']['kernel'], d['roofline']['frac'])
PY
done
