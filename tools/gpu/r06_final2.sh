# round 6, last state of the tree: the whole GPU suite, the LZ line again (the LZ kernels moved after r06_final.sh ran: emit, mark, the list of moved words, relink), its trace and its multi-device curve
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_final2; mkdir -p $O
python bench.py --config lz > $O/config_lz_bench.json 2> $O/lz.err
rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_lz.json 2> $O/prof_lz.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py $DB $O/config3_lz_kernel_stats.md > /dev/null 2>&1; rm -rf $O/prof
python bench.py --config lz --in-process-devices 1,2,3,4,8 > $O/multi_device_logical_lz.json 2> $O/md3.err
python tools/gpu/lz_rounds.py 2>&1 | grep -v amdgpu.ids | grep -i "rounds\|parse" > $O/lz_rounds.txt
timeout 200 python tools/gpu/lz_seg_fuzz.py 120 600 > $O/lz_seg_fuzz.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tail -1 $O/lz_seg_fuzz.log; tail -3 $O/gpu_suite.log; cut -c1-200 $O/config_lz_bench.json
