# round 6: idle stretches of the last encode + decode of the default configuration (host syncs between kernels)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_gaps; mkdir -p $O
for cfg in bwt lz; do
rocprofv3 --kernel-trace -d $O/prof_$cfg -- python bench.py --config $cfg --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_$cfg.json 2> $O/prof_$cfg.err
DB=$(find $O/prof_$cfg -name "*.db" | head -1)
if [ $cfg = bwt ]; then python tools/step_gaps.py $DB knz_ss_hist 60 > $O/gaps_$cfg.txt 2>&1; else python tools/step_gaps.py $DB knz_lz_keys 60 > $O/gaps_$cfg.txt 2>&1; fi
rm -rf $O/prof_$cfg; echo "== $cfg"; tail -40 $O/gaps_$cfg.txt | cut -c1-200
done
