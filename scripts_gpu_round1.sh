#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5) > gpurun_out/pytest_gpu.log
(timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/bench.err | tail -1) > gpurun_out/bench.json
cat gpurun_out/bench.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/pmc_huf -o r1 -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $GRAFT_REPO_ROOT/gpurun_out/pmc_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/pmc.err)
tail -3 gpurun_out/pytest_gpu.log
