#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "BWT or bwt or config4 or full_size or list_ranking" --durations=5 > gpurun_out/s_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/s_pytest.log
for m in own rocprim; do
  if [ $m = rocprim ]; then export KNZ_PRIMS=rocprim; else unset KNZ_PRIMS; fi
  timeout 600 python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > gpurun_out/s_bench_$m.json 2> gpurun_out/s_bench_$m.err; echo "bench $m rc=$?"
  python - <<PY
import json
d=json.load(open('gpurun_out/s_bench_$m.json'))
print('$m', d['value'], d['encode_MBps'], d['decode_MBps'], d['roofline']['all_stage_ms'], d['bit_exact_vs_oracle'])
PY
done
unset KNZ_PRIMS
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_s -o kt -- python bench.py --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/s_prof.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_s -name '*.db' | head -1) gpurun_out/s_kernel_stats.md > /dev/null; echo "stats rc=$?"; head -24 gpurun_out/s_kernel_stats.md
find gpurun_out -name '*.db' -size +8M -delete
