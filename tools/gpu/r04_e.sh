#!/bin/bash
# round 4: after run-aware keys: tests, variants, trace of one variant ($1, default t2048_512)
mkdir -p gpurun_out; export TMPDIR=/tmp
TV=${1:-t2048_512}
timeout 900 python -m pytest tests -m gpu -x -q -k "transform_objects or config4 or stress or fuzz or bwt or l5 or stream_bit_exact" > gpurun_out/e_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/e_pytest.log
for v in default t2048_512 t1024_256 t1024_512; do
lib=$PWD/kanzi-go_amd/variants/libknz_$v.so; [ $v = default ] && lib=$PWD/kanzi-go_amd/libknz_gpu.so
KNZ_BWT_PROF=1 KNZ_GPU_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --steps 4 --warmup 1 > gpurun_out/e_bench_$v.json 2> gpurun_out/e_bench_$v.err; echo "$v rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/e_bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['encode_MBps'], d['bit_exact_vs_oracle'], d['roofline'].get('phase_ms_per_step'), d['roofline']['all_stage_ms']['enc_transform'])
PY
done
grep "suffix sort" gpurun_out/e_bench_default.err | head -12
rm -rf gpurun_out/prof_kt
KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_$TV.so timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/e_prof_kt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) gpurun_out/e_kernel_stats_config4.md > /dev/null; echo "stats rc=$?"
