#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --config lz --steps 3 --warmup 1 > gpurun_out/z_bench_lz.json 2> gpurun_out/z_bench_lz.err; echo "lz rc=$?"
rm -rf gpurun_out/prof_kt_lz
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt_lz -o kt -- python bench.py --config lz --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/z_prof_kt_lz.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt_lz -name '*.db' | head -1) gpurun_out/z_kernel_stats_config3_lz.md > /dev/null; echo "stats lz rc=$?"
find gpurun_out -name '*.db' -size +8M -delete
KNZ_LZS_PROF=1 timeout 600 python tools/gpu/lz_rounds.py > gpurun_out/ab_lz_rounds.txt 2>&1
grep "parse_kernel\|rounds per block\|stream bytes" gpurun_out/ab_lz_rounds.txt | tail -8 | cut -c1-400
python - <<'PY'
import json
d=json.loads(open('gpurun_out/z_bench_lz.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['encode_MBps'], d['decode_MBps'], r['kernel'], r['frac'], r.get('avg_launch_ms'), r.get('traffic_over_algorithmic'), d.get('cpu_baseline',{}).get('encode_MBps'), d.get('cpu_baseline',{}).get('decode_MBps'), d['bit_exact_vs_oracle'], d.get('fallback_counters_last_batch'), r['all_stage_ms'], d.get('host_hook_MBps'))
PY
