#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o r1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/pmc_fetch.json 2> $R/gpurun_out/pmc_fetch.err
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o r1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $R/gpurun_out/pmc_write.json 2> $R/gpurun_out/pmc_write.err
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bwt -o r1 -- python $R/bench.py --config bwt --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench_bwt.json 2> $R/gpurun_out/prof_bwt.err
cd $R
timeout 900 python bench.py --config bwt --steps 2 --warmup 1 > gpurun_out/bench_bwt.json 2> gpurun_out/bench_bwt.err
timeout 600 python bench.py --config ans0 --steps 3 --warmup 1 > gpurun_out/bench_ans0.json 2> gpurun_out/bench_ans0.err
timeout 900 python bench.py --config lz --steps 1 --warmup 1 > gpurun_out/bench_lz.json 2> gpurun_out/bench_lz.err
for f in bench_bwt bench_ans0 bench_lz; do python - <<PY
import json
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['encode_MBps'], d['decode_MBps'], d.get('cpu_baseline',{}).get('encode_MBps'), d.get('cpu_baseline',{}).get('decode_MBps'))
PY
done
