# round 6: kernel timeline of one decode of the -l 5 preset
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_l5; mkdir -p $O
rocprofv3 --kernel-trace -d $O/prof -- python bench.py --config l5 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof.json 2> $O/prof.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/decode_timeline.py $DB 500 knz_ans0_walk_decode > $O/decode_timeline.txt 2>&1; rm -rf $O/prof; cat $O/decode_timeline.txt | cut -c1-140 | tail -40
