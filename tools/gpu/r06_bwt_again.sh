# round 6: the default bench line three more times on the last tree (the chains' speed moves by a few per cent from box to box and run to run)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_bwt_again; mkdir -p $O
for k in 1 2 3; do python bench.py --no-cpu-baseline --no-pmc --no-host-hook --steps 10 --warmup 3 > $O/bwt_$k.json 2> $O/bwt_$k.err; python -c "
import json,sys
d=json.loads(open('$O/bwt_$k.json').read().strip().splitlines()[-1]); print('run $k value', d['value'], 'enc', d['encode_MBps'], 'dec', d['decode_MBps'], d['roofline']['all_stage_ms'])"; done
