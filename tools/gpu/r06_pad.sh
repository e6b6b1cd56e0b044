# round 6 experiment (NOT in the tree: the launches pass 0 again): do the fused chain's waves share compute units / SIMDs with other waves? Dynamic LDS on its launches keeps its workgroups apart
# (60 KB: at most two per compute unit and none beside a decoder workgroup; 20 KB: none beside a decoder workgroup). Variants: a KNZ_PIPE_LDS_PAD macro as the dynamic LDS of the two knz_zrlti_rank_pipe_kernel launches, bash tools/gpu/run.sh "lib:pad60:-DKNZ_PIPE_LDS_PAD=61440".
# Result: no. decode entropy + transform ms, two runs of 10 steps each: base 518.6 / 523.4, 60 KB 524.8 / 547.8, 20 KB 523.0 / 541.8 - the chains move by 2-5 % from run to run either way.
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_pad; mkdir -p $O
for v in base pad60 pad20 base pad60 pad20; do
  if [ $v = base ]; then unset KNZ_GPU_LIB; else export KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_$v.so; fi
  python bench.py --no-cpu-baseline --no-pmc --no-host-hook --steps 10 --warmup 3 > $O/$v.json 2> $O/$v.err; python -c "
import json,sys
d=json.loads(open('$O/$v.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$v value', d['value'], 'dec', d['decode_MBps'], 'exact', d.get('bit_exact_vs_oracle'), r['all_stage_ms']['dec_entropy'], r['all_stage_ms']['dec_transform'], round(r['all_stage_ms']['dec_entropy'] + r['all_stage_ms']['dec_transform'], 1))"; done
