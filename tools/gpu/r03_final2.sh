#!/bin/bash
# evidence refresh after the last inverse RANK change: the GPU suite, the default bench line with its PMC passes, its kernel trace, the -l 5 line
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/z_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/z_pytest.log
timeout 1200 python bench.py > gpurun_out/z_bench.json 2> gpurun_out/z_bench.err; echo "bench rc=$?"
rm -rf gpurun_out/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/z_prof_kt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) gpurun_out/z_kernel_stats_config4.md > /dev/null; echo "stats rc=$?"
find gpurun_out -name '*.db' -size +8M -delete
timeout 900 python bench.py --config l5 --steps 3 --warmup 1 > gpurun_out/z_bench_l5.json 2> gpurun_out/z_bench_l5.err; echo "l5 rc=$?"
KNZ_RANK_PROF=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > /dev/null 2> gpurun_out/z_rank_prof.err; grep "inverse RANK chain" gpurun_out/z_rank_prof.err | tail -1
python - <<'PY'
import json
for n in ['z_bench','z_bench_l5']:
    d=json.loads(open(f'gpurun_out/{n}.json').read().strip().splitlines()[-1]); r=d['roofline']; c=d['cpu_baseline']; h=d.get('host_hook_MBps') or {}
    print(n, d['value'], d['encode_MBps'], d['decode_MBps'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r.get('traffic_over_algorithmic'), c.get('encode_MBps'), c.get('decode_MBps'), h.get('encode'), h.get('decode'), r['all_stage_ms'])
PY
