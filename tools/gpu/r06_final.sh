# round 6, final state (one gpurun call): all bench configurations, kernel traces, the in-process multi-device curves, fuzz with fresh seeds, the whole GPU suite
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_final; mkdir -p $O
python bench.py > $O/config_bwt_bench.json 2> $O/bwt.err
python bench.py --config l5 > $O/config_l5_bench.json 2> $O/l5.err
python bench.py --config lz > $O/config_lz_bench.json 2> $O/lz.err
python bench.py --config huffman > $O/config_huffman_bench.json 2> $O/huf.err
python bench.py --config ans0 > $O/config_ans0_bench.json 2> $O/ans0.err
rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_bench.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py $DB $O/config4_kernel_stats.md > /dev/null 2>&1; rm -rf $O/prof
rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_lz.json 2> $O/prof_lz.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py $DB $O/config3_lz_kernel_stats.md > /dev/null 2>&1; rm -rf $O/prof
python bench.py --config bwt --in-process-devices 1,2,4,8 --depth 104 > $O/multi_device_logical_bwt_depth104.json 2> $O/md1.err
python bench.py --config bwt --in-process-devices 1,2,3,4,8 > $O/multi_device_logical_bwt.json 2> $O/md2.err
python bench.py --config lz --in-process-devices 1,2,3,4,8 > $O/multi_device_logical_lz.json 2> $O/md3.err
python bench.py --config huffman --in-process-devices 1,2,3,4,8 > $O/multi_device_logical_huffman.json 2> $O/md4.err
timeout 300 python tools/gpu/multi_handle_check.py 6 2 > $O/multi_handle_check.log 2>&1
timeout 400 python tools/gpu/ext_fuzz.py 240 7000 > $O/ext_fuzz.log 2>&1
timeout 300 python tools/gpu/lz_order_check.py 500 > $O/lz_order_check.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/gpu_suite.log
tail -2 $O/ext_fuzz.log; tail -1 $O/lz_order_check.log; tail -2 $O/multi_handle_check.log; tail -3 $O/gpu_suite.log; for f in bwt l5 lz huffman ans0; do cut -c1-170 $O/config_${f}_bench.json; done
