# round 6: where the decode of configs[3] spends its wall time - kernel timeline of the last decode of a traced bench run
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_timeline; mkdir -p $O
rocprofv3 --kernel-trace -d $O/prof -- python bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_bwt.json 2> $O/prof_bwt.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/decode_timeline.py $DB 300 > $O/decode_timeline.txt 2>&1; rm -rf $O/prof; cat $O/decode_timeline.txt | tail -70
