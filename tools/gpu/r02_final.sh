#!/bin/bash
# round 2 evidence: the whole GPU suite, the default bench line (configs[3], live PMC passes, host hook, CPU baseline), the kernel trace of
# the same command, the other BASELINE configs' bench lines
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/f_pytest.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/f_pytest.log
timeout 900 python bench.py > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err; echo "bench rc=$?"; cat gpurun_out/f_bench.json; tail -3 gpurun_out/f_bench.err
rm -rf gpurun_out/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/f_prof_kt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) gpurun_out/f_kernel_stats_config4.md > /dev/null; echo "stats rc=$?"
find gpurun_out -name '*.db' -size +8M -delete
for c in huffman ans0 lz; do
  timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-pmc > gpurun_out/f_bench_$c.json 2> gpurun_out/f_bench_$c.err; echo "$c rc=$?"
done
timeout 900 python bench.py --config fpaq --steps 1 --warmup 0 --no-pmc --no-host-hook > gpurun_out/f_bench_fpaq.json 2> gpurun_out/f_bench_fpaq.err; echo "fpaq rc=$?"
python - <<'PY'
import json
for n in ['huffman','ans0','lz','fpaq']:
    try:
        d=json.loads(open(f'gpurun_out/f_bench_{n}.json').read().strip().splitlines()[-1])
        print(n, d['value'], d['encode_MBps'], d['decode_MBps'], d['roofline']['kernel'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('encode_MBps'), d.get('cpu_baseline',{}).get('decode_MBps'), d['bit_exact_vs_oracle'], d.get('host_hook_MBps'))
    except Exception as e: print(n,'ERR',e)
PY
