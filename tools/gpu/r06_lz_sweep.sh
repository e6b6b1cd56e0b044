cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for seg in 512 768 1024 1536 2048; do
  KNZ_LZ_SEG=$seg timeout 600 python bench.py --config lz --no-pmc --no-cpu-baseline --no-host-hook --steps 4 --warmup 1 > gpurun_out/x.json 2> gpurun_out/x.err
  python - gpurun_out/x.json $seg <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; k = r.get("kernel_ms_per_step", {})
print("seg", sys.argv[2], "enc", d["encode_MBps"], "enc_transform", r["all_stage_ms"]["enc_transform"], "parse", k.get("knz_lzs_parse_lanes_kernel"), "rounds", d["fallback_counters_last_batch"]["lz_forward_rounds"], d.get("bit_exact_vs_oracle"))
PY
done
