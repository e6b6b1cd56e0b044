# round 6 experiment: the splitters of the inverse BWT's list ranking on every 127th / 61st / 257th slot instead of every 128th (do the walks resonate with power-of-two record sizes?) Variants are built first with: bash tools/gpu/run.sh "lib:split127:-DKNZ_BWT_SPLIT=127u" ... (kanzi-go_amd/variants/, not tracked). Result: profiles/README.md, bwt.hip.
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_split; mkdir -p $O
for v in base split127 split191 split257 split383 split509 base split257; do
  if [ $v = base ]; then unset KNZ_GPU_LIB; else export KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_$v.so; fi
  timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-host-hook --steps 5 --warmup 1 > $O/$v.json 2> $O/$v.err
  python - $O/$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; k = r["kernel_ms_per_step"]
print(sys.argv[2], "dec", d["decode_MBps"], "ms", r.get("decode_ms"), "exact", d.get("bit_exact_vs_oracle"), "dec_transform", r["all_stage_ms"]["dec_transform"], {n: v for n, v in k.items() if "bwt_inv" in n})
PY
done
