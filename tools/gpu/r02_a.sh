#!/bin/bash
# round 2, run A: inverse RANK chain variants (A/B), the tests around them, config 4 bench line
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k "rank_chain or transform_objects or config4 or (stream_bit_exact and RANK)" > gpurun_out/a_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/a_pytest.log
timeout 600 python tools/gpu/rank_variants.py > gpurun_out/a_rank_variants.json 2> gpurun_out/a_rank_variants.err; echo "variants rc=$?"; cat gpurun_out/a_rank_variants.json; tail -3 gpurun_out/a_rank_variants.err
timeout 600 python bench.py --config bwt --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/a_bench_bwt.json 2> gpurun_out/a_bench_bwt.err; echo "bench rc=$?"; cat gpurun_out/a_bench_bwt.json; tail -3 gpurun_out/a_bench_bwt.err
