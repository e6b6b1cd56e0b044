# round 6, last state of the tree (after the LZ emit / mark / relink and inverse-chain changes): the extended fuzz, the stress checks and the other four bench lines again
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_final3; mkdir -p $O
timeout 400 python tools/gpu/ext_fuzz.py 300 8000 > $O/ext_fuzz.log 2>&1
timeout 300 python tools/gpu/lz_order_check.py 500 > $O/lz_order_check.log 2>&1
timeout 300 python tools/gpu/multi_handle_check.py 6 2 > $O/multi_handle_check.log 2>&1
for cfg in bwt l5 huffman ans0; do timeout 1500 python bench.py --config $cfg > $O/config_${cfg}_bench.json 2> $O/$cfg.err; echo "$cfg rc=$?"; done
rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_bwt.json 2> $O/prof_bwt.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py $DB $O/config4_kernel_stats.md > /dev/null 2>&1; rm -rf $O/prof
tail -2 $O/ext_fuzz.log; tail -1 $O/lz_order_check.log; tail -2 $O/multi_handle_check.log; for f in bwt l5 huffman ans0; do cut -c1-170 $O/config_${f}_bench.json; done
