cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_lz; mkdir -p $O
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
k = r.get("kernel_ms_per_step", {}); l = r.get("kernel_launches_per_step", {})
print(sys.argv[2], "| enc", d["encode_MBps"], "exact", d.get("bit_exact_vs_oracle"), d.get("bit_exact_vs_reference"), "| enc_transform ms", r["all_stage_ms"]["enc_transform"],
      "| parse", k.get("knz_lzs_parse_lanes_kernel", k.get("knz_lzs_parse_kernel")), "launches", l.get("knz_lzs_parse_lanes_kernel", l.get("knz_lzs_parse_kernel")), "| fallbacks", d.get("fallback_counters_last_batch"))
PY
}
for v in default nowin cap32k nowin_cap32k; do
 for seg in 384 512 768; do
  if [ $v != default ]; then export KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_$v.so; else unset KNZ_GPU_LIB; fi
  KNZ_LZ_SEG=$seg timeout 600 python bench.py --config lz --no-pmc --no-cpu-baseline --no-host-hook --steps 3 --warmup 1 > $O/x.json 2> $O/x.err; show $O/x.json "$v seg $seg"
 done
done
