# round 6, the LAST tree again (after the chain's streams got priorities of their own): the suite, the five bench lines, the default line's trace + timeline + gaps,
# per-decode times through the handle's own stream, the multi-device curves of the default configuration, smoke
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_last2; mkdir -p $O
python bench.py > $O/config_bwt_bench.json 2> $O/bwt.err
python bench.py --config l5 > $O/config_l5_bench.json 2> $O/l5.err
python bench.py --config lz > $O/config_lz_bench.json 2> $O/lz.err
python bench.py --config huffman > $O/config_huffman_bench.json 2> $O/huf.err
python bench.py --config ans0 > $O/config_ans0_bench.json 2> $O/ans0.err
rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_bench.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py $DB $O/config4_kernel_stats.md > /dev/null 2>&1; python tools/decode_timeline.py $DB 300 > $O/decode_timeline.txt 2>&1; python tools/step_gaps.py $DB knz_ss_hist 60 > $O/step_gaps.txt 2>&1; rm -rf $O/prof
python tools/gpu/decode_steps.py 20 2>&1 | grep -v amdgpu.ids | tail -2 > $O/decode_steps.txt
python bench.py --config bwt --in-process-devices 1,2,4,8 --depth 104 > $O/multi_device_logical_bwt_depth104.json 2> $O/md1.err
python bench.py --config bwt --in-process-devices 1,2,3,4,8 > $O/multi_device_logical_bwt.json 2> $O/md2.err
timeout 300 python tools/gpu/multi_handle_check.py 6 2 2>&1 | grep -v amdgpu.ids > $O/multi_handle_check.log
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $O/smoke.log
cat $O/smoke.log; cat $O/decode_steps.txt; tail -2 $O/multi_handle_check.log; tail -3 $O/gpu_suite.log; for f in bwt l5 lz huffman ans0; do cut -c1-190 $O/config_${f}_bench.json; done
