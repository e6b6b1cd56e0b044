cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_lz; mkdir -p $O
for seg in 1024 2048; do echo "== lanes seg $seg"; KNZ_LZ_SEG=$seg python tools/gpu/lz_rounds.py 2>&1 | grep -v amdgpu.ids | grep -i "rounds\|parse\|emit\|relink\|compare\|init\|cand\|cp_kernel\|keys\|scatter\|hist" ; done
rm -rf $O/prof; rocprofv3 --kernel-trace --stats -d $O/prof -o kt -- python bench.py --config lz --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > $O/trace.log 2>&1
db=$(find $O/prof -name '*.db' | head -1); python tools/rocpd_summary.py $db $O/r06_lz_lanes_kernel_stats_v1.md > /dev/null; sed -n 1,30p $O/r06_lz_lanes_kernel_stats_v1.md | cut -c1-150
find $O -name '*.db' -delete
