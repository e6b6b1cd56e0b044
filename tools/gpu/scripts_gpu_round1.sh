#!/bin/bash
# full check: GPU parity suite, default bench line (with cpu_baseline), kernel trace of the same command
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
timeout 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo "bench rc=$?"; tail -c 2500 gpurun_out/bench_full.json
rm -rf gpurun_out/prof_kt
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/prof_kt.log 2>&1; echo "rocprof rc=$?"
f=$(find gpurun_out/prof_kt -name '*.db' | head -1); [ -n "$f" ] && python tools/rocpd_summary.py $f gpurun_out/kernel_stats.md | head -20
find gpurun_out/prof_kt -name '*.db' -size +20M -delete
