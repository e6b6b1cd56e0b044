# round 6: where every block's fused ZRLT / RANK chain spends its time under the rANS-1 decoder (measure build, KNZ_RANK_PROF)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_pipe_prof; mkdir -p $O
export KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_measure.so
KNZ_RANK_PROF=1 timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --no-verify --steps 1 --warmup 1 > $O/quick.json 2> $O/quick.err
grep -A30 "fused ZRLT/RANK inverse" $O/quick.err | tail -32
