#!/bin/bash
# round 2, run B: the whole GPU suite, the default bench line (configs[3]) with its live PMC passes, kernel trace of the same command
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/b_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/b_pytest.log
timeout 900 python bench.py > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err; echo "bench rc=$?"; cat gpurun_out/b_bench.json; tail -3 gpurun_out/b_bench.err
rm -rf gpurun_out/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/b_prof_kt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) gpurun_out/b_kernel_stats_config4.md > /dev/null; echo "stats rc=$?"
find gpurun_out -name '*.db' -size +8M -delete
