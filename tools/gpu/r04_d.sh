#!/bin/bash
# round 4: sg_sort phase cycles, T / THREADS variants, kernel trace of the default (after the keys kernel split)
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "transform_objects or config4 or stress or fuzz or bwt" > gpurun_out/d_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/d_pytest.log
for v in prof1024_512 prof2048_256; do
KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_$v.so KNZ_BWT_PROF=1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --no-verify --steps 1 --warmup 1 > gpurun_out/d_bench_$v.json 2> gpurun_out/d_bench_$v.err; echo "$v rc=$?"
grep "sg_sort" gpurun_out/d_bench_$v.err | head -4
done
for v in default t2048_512 t1024_256 t1024_512; do
lib=$PWD/kanzi-go_amd/variants/libknz_$v.so; [ $v = default ] && lib=$PWD/kanzi-go_amd/libknz_gpu.so
KNZ_GPU_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --steps 4 --warmup 1 > gpurun_out/d_bench_$v.json 2> gpurun_out/d_bench_$v.err; echo "$v rc=$?"
python - <<PY
import json
d=json.loads(open('gpurun_out/d_bench_$v.json').read().strip().splitlines()[-1])
print('$v', d['encode_MBps'], d['bit_exact_vs_oracle'], d['roofline'].get('phase_ms_per_step'), d['roofline']['all_stage_ms']['enc_transform'])
PY
done
rm -rf gpurun_out/prof_kt
KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_t1024_512.so timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/d_prof_kt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) gpurun_out/d_kernel_stats_config4.md > /dev/null; echo "stats rc=$?"
sed -n 3,30p gpurun_out/d_kernel_stats_config4.md
