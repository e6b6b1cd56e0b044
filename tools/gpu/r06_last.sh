# round 6, the LAST tree (one gpurun call): every bench configuration, the two kernel traces, the decode timeline, the in-process multi-device curves, fuzz with
# fresh seeds, the stress checks, the whole GPU suite, smoke
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_last; mkdir -p $O
python bench.py > $O/config_bwt_bench.json 2> $O/bwt.err
python bench.py --config l5 > $O/config_l5_bench.json 2> $O/l5.err
python bench.py --config lz > $O/config_lz_bench.json 2> $O/lz.err
python bench.py --config huffman > $O/config_huffman_bench.json 2> $O/huf.err
python bench.py --config ans0 > $O/config_ans0_bench.json 2> $O/ans0.err
rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_bench.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py $DB $O/config4_kernel_stats.md > /dev/null 2>&1; python tools/decode_timeline.py $DB 300 > $O/decode_timeline.txt 2>&1; python tools/step_gaps.py $DB knz_ss_hist 60 > $O/step_gaps.txt 2>&1; rm -rf $O/prof
rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_lz.json 2> $O/prof_lz.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py $DB $O/config3_lz_kernel_stats.md > /dev/null 2>&1; rm -rf $O/prof
python tools/gpu/lz_rounds.py 2>&1 | grep -v amdgpu.ids | grep -i "rounds\|parse" > $O/lz_rounds.txt
python bench.py --config bwt --in-process-devices 1,2,4,8 --depth 104 > $O/multi_device_logical_bwt_depth104.json 2> $O/md1.err
python bench.py --config bwt --in-process-devices 1,2,3,4,8 > $O/multi_device_logical_bwt.json 2> $O/md2.err
python bench.py --config lz --in-process-devices 1,2,3,4,8 > $O/multi_device_logical_lz.json 2> $O/md3.err
python bench.py --config huffman --in-process-devices 1,2,3,4,8 > $O/multi_device_logical_huffman.json 2> $O/md4.err
timeout 300 python tools/gpu/multi_handle_check.py 6 2 2>&1 | grep -v amdgpu.ids > $O/multi_handle_check.log
timeout 330 python tools/gpu/ext_fuzz.py 240 10000 > $O/ext_fuzz.log 2>&1
timeout 300 python tools/gpu/lz_order_check.py 500 2>&1 | grep -v amdgpu.ids > $O/lz_order_check.log
timeout 200 python tools/gpu/lz_seg_fuzz.py 150 700 > $O/lz_seg_fuzz.log 2>&1
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.log
cat $O/smoke.log; tail -1 $O/ext_fuzz.log; tail -1 $O/lz_order_check.log; tail -1 $O/lz_seg_fuzz.log; tail -2 $O/multi_handle_check.log; tail -3 $O/gpu_suite.log; for f in bwt l5 lz huffman ans0; do cut -c1-170 $O/config_${f}_bench.json; done
