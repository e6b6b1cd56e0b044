# round 6, after the splitters of the inverse BWT moved to a prime distance: the whole GPU suite, the two bench lines with a BWT in them, the trace, the fuzz
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_final4; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/gpu_suite.log; tail -3 $O/gpu_suite.log
timeout 300 python tools/gpu/ext_fuzz.py 200 9000 > $O/ext_fuzz.log 2>&1; tail -1 $O/ext_fuzz.log
for cfg in bwt l5; do timeout 1500 python bench.py --config $cfg > $O/config_${cfg}_bench.json 2> $O/$cfg.err; echo "$cfg rc=$?"; done
rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_bwt.json 2> $O/prof_bwt.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py $DB $O/config4_kernel_stats.md > /dev/null 2>&1; python tools/decode_timeline.py $DB 300 > $O/decode_timeline.txt 2>&1; rm -rf $O/prof
python bench.py --in-process-devices 1,2,4,8 --depth 104 > $O/multi_device_logical_bwt_depth104.json 2> $O/md.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for f in bwt l5; do cut -c1-170 $O/config_${f}_bench.json; done; tail -22 $O/decode_timeline.txt
