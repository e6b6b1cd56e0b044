# round 6 experiment: passes and hops of the pointer-jumping kernel in front of the LZ inverse's gather. Variants are built first with
# bash tools/gpu/run.sh "lib:p1h8:-DKNZ_LZI_JUMP_PASSES=1" ... (kanzi-go_amd/variants/, not tracked)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_jump; mkdir -p $O
for v in base p2h2 p3h1 p3h2 p4h1 p4h2 p5h1 p3h2; do
  if [ $v = base ]; then unset KNZ_GPU_LIB; else export KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_$v.so; fi
  timeout 600 python bench.py --config lz --no-cpu-baseline --no-pmc --no-host-hook --steps 5 --warmup 2 > $O/$v.json 2> $O/$v.err
  python - $O/$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; k = r["kernel_ms_per_step"]
print(sys.argv[2], "dec", d["decode_MBps"], "exact", d.get("bit_exact_vs_oracle"), "dec_transform", r["all_stage_ms"]["dec_transform"], {n: v for n, v in k.items() if "lzi" in n})
PY
done
