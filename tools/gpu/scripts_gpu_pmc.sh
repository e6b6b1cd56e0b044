#!/bin/bash
# HBM traffic of the bench kernels: two PMC passes (own runs, kernel trace only), then the kernel trace summary
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/prof_kt
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify ${BENCH_ARGS} > gpurun_out/pmc_f.log 2>&1; echo "fetch rc=$?"
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify ${BENCH_ARGS} > gpurun_out/pmc_w.log 2>&1; echo "write rc=$?"
python tools/pmc_traffic.py $(find gpurun_out/pmc_f -name '*.db' | head -1) $(find gpurun_out/pmc_w -name '*.db' | head -1) gpurun_out/pmc_traffic.json | tail -30
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 ${BENCH_ARGS} > gpurun_out/prof_kt.log 2>&1; echo "rocprof rc=$?"
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) gpurun_out/kernel_stats.md | head -16
find gpurun_out -name '*.db' -size +8M -delete
