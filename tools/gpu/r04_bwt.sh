#!/bin/bash
# round 4, suffix sort (bwt_sort.hip): the BWT-related GPU tests, the default bench line without the slow extras, per-round diagnostics,
# and the kernel trace of the same command
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -k "transform_objects or config4 or stream_bit_exact or stress or block_batch or fuzz or bwt or l5" --durations=5 > gpurun_out/a_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/a_pytest.log
KNZ_BWT_PROF=1 timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --steps 5 --warmup 2 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "bench rc=$?"
grep "suffix sort" gpurun_out/a_bench.err | head -24
python - <<'PY'
import json
d=json.loads(open('gpurun_out/a_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['encode_MBps'], d['decode_MBps'], d['bit_exact_vs_oracle'])
print(d['roofline'].get('all_stage_ms'))
print(d['roofline'].get('kernel_ms_per_step'))
PY
rm -rf gpurun_out/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/a_prof_kt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) gpurun_out/a_kernel_stats_config4.md > /dev/null; echo "stats rc=$?"
head -40 gpurun_out/a_kernel_stats_config4.md
find gpurun_out -name '*.db' -size +8M -delete
