# round 6: the fused chain's two streams at priorities of their own (hardware queues per priority level): a decode of configs[3] through the handle's own stream
# in a torch process (was 833 ms: the two chain launches shared a queue), the bench line, the decode timeline
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_prio; mkdir -p $O
python tools/gpu/decode_steps.py 12 2>&1 | grep -v amdgpu.ids | tail -2
for k in 1 2; do python bench.py --no-cpu-baseline --no-pmc --no-host-hook > $O/bwt_$k.json 2> $O/bwt_$k.err; python -c "
import json,sys
d=json.loads(open('$O/bwt_$k.json').read().strip().splitlines()[-1]); print('run $k value', d['value'], 'enc', d['encode_MBps'], 'dec', d['decode_MBps'], d.get('bit_exact_vs_oracle'), d['roofline']['all_stage_ms'])"; done
rocprofv3 --kernel-trace -d $O/prof -- python bench.py --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_bwt.json 2> $O/prof_bwt.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/decode_timeline.py $DB 300 > $O/decode_timeline.txt 2>&1; rm -rf $O/prof; head -5 $O/decode_timeline.txt | cut -c1-130; grep "bwt_inv_keys\|decode span" $O/decode_timeline.txt | cut -c1-130
timeout 900 python -m pytest tests -m gpu -x -q -k "rank_pipe or full_size_config4 or multi_device or deep_batches or go_shim" 2>&1 | tail -2
