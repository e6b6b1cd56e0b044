#!/bin/bash
# round 4: hand-written ANS1 encode loop + tiled forward RANK: tests, default bench, A/B of the encode loop, trace
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "ans1 or ANS1 or rank or RANK or transform_objects or config4 or stress or fuzz or bwt or l5 or stream_bit_exact or mtft or srt or golden" > gpurun_out/f_pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/f_pytest.log
for v in default plain; do
[ $v = plain ] && export KNZ_ANS1_ENC_PLAIN=1
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --steps 5 --warmup 2 > gpurun_out/f_bench_$v.json 2> gpurun_out/f_bench_$v.err; echo "$v rc=$?"
unset KNZ_ANS1_ENC_PLAIN
python - <<PY
import json
d=json.loads(open('gpurun_out/f_bench_$v.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$v', d['value'], d['encode_MBps'], d['decode_MBps'], d['bit_exact_vs_oracle'], r.get('phase_ms_per_step'), r['all_stage_ms'])
print({k:v for k,v in r['kernel_ms_per_step'].items()})
PY
done
rm -rf gpurun_out/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/f_prof_kt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) gpurun_out/f_kernel_stats_config4.md > /dev/null; echo "stats rc=$?"
python tools/trace_rounds.py
