cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_lz; mkdir -p $O
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
k = r.get("kernel_ms_per_step", {}); l = r.get("kernel_launches_per_step", {})
print(sys.argv[2], "| enc", d["encode_MBps"], "dec", d["decode_MBps"], "exact", d.get("bit_exact_vs_oracle"), d.get("bit_exact_vs_reference"), "| enc_transform ms", r["all_stage_ms"]["enc_transform"],
      "| parse", k.get("knz_lzs_parse_lanes_kernel", k.get("knz_lzs_parse_kernel")), "launches", l.get("knz_lzs_parse_lanes_kernel", l.get("knz_lzs_parse_kernel")), "| fallbacks", d.get("fallback_counters_last_batch"))
PY
}
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "lz_forward_forms or lz_streams_small or full_size_config3 or lz_first" > $O/pytest_lz.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_lz.log
for seg in 512 1024 2048; do
  KNZ_LZ_SEG=$seg timeout 600 python bench.py --config lz --no-pmc --no-cpu-baseline --no-host-hook --steps 3 --warmup 1 > $O/lanes_$seg.json 2> $O/lanes_$seg.err; show $O/lanes_$seg.json "lanes $seg"
done
for seg in 1024; do echo "== lanes seg $seg"; KNZ_LZ_SEG=$seg python tools/gpu/lz_rounds.py 2>&1 | grep -v amdgpu.ids | grep -i "rounds\|parse\|emit\|relink\|compare" ; done
rm -rf $O/prof; rocprofv3 --kernel-trace --stats -d $O/prof -o kt -- python bench.py --config lz --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > $O/trace.log 2>&1
db=$(find $O/prof -name '*.db' | head -1); python tools/rocpd_summary.py $db $O/r06_lz_lanes_kernel_stats_v2.md > /dev/null; sed -n 1,16p $O/r06_lz_lanes_kernel_stats_v2.md | cut -c1-150
find $O -name '*.db' -delete
