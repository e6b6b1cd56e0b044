#!/bin/bash
# round 3 evidence: the whole GPU suite, the default bench line (configs[3], live PMC passes, host hook, CPU baseline), the kernel trace of the
# same command, the other BASELINE configs' bench lines WITH their PMC passes, configs[4] at its stated size (10^9 bytes, 30 blocks)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/z_pytest.log 2>&1; echo "pytest rc=$?"; tail -14 gpurun_out/z_pytest.log
timeout 1200 python bench.py > gpurun_out/z_bench.json 2> gpurun_out/z_bench.err; echo "bench rc=$?"; cat gpurun_out/z_bench.json; tail -3 gpurun_out/z_bench.err
rm -rf gpurun_out/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/z_prof_kt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) gpurun_out/z_kernel_stats_config4.md > /dev/null; echo "stats rc=$?"
rm -rf gpurun_out/prof_kt_lz
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt_lz -o kt -- python bench.py --config lz --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/z_prof_kt_lz.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt_lz -name '*.db' | head -1) gpurun_out/z_kernel_stats_config3_lz.md > /dev/null; echo "stats lz rc=$?"
find gpurun_out -name '*.db' -size +8M -delete
for c in lz huffman ans0 l5; do
  timeout 900 python bench.py --config $c --steps 3 --warmup 1 > gpurun_out/z_bench_$c.json 2> gpurun_out/z_bench_$c.err; echo "$c rc=$?"
done
timeout 2400 python bench.py --config fpaq --steps 1 --warmup 0 --no-host-hook > gpurun_out/z_bench_fpaq.json 2> gpurun_out/z_bench_fpaq.err; echo "fpaq rc=$?"
python - <<'PY'
import json
for n in ['lz','huffman','ans0','l5','fpaq']:
    try:
        d=json.loads(open(f'gpurun_out/z_bench_{n}.json').read().strip().splitlines()[-1])
        r=d['roofline']
        print(n, d['value'], d['encode_MBps'], d['decode_MBps'], r['kernel'], r['frac'], r.get('traffic_over_algorithmic'), d.get('cpu_baseline',{}).get('encode_MBps'), d.get('cpu_baseline',{}).get('decode_MBps'), d['bit_exact_vs_oracle'], d.get('fallback_counters_last_batch'))
    except Exception as e: print(n,'ERR',e)
PY
