#!/bin/bash
# One script for the GPU box (gpurun -- 'bash tools/gpu/run.sh <step> ...'); every step writes under gpurun_out/ with the prefix $TAG (default r04).
#   tests[:EXPR]      pytest -m gpu (EXPR = -k expression)
#   bench[:CONFIG]    bench.py line of a config (bwt = default, lz, huffman, ans0, l5, fpaq) with its live PMC passes, host hook, CPU baseline
#   quick[:CONFIG]    bench.py without the slow extras (no CPU baseline / PMC / host hook), 5 steps
#   trace[:CONFIG]    rocprofv3 --kernel-trace --stats of the quick command -> <TAG>_kernel_stats_<config>.md (+ the suffix sort's per-round timeline)
#   lib:NAME:FLAGS    (run on the build side) hipcc build of kanzi-go_amd/variants/libknz_NAME.so with extra FLAGS, e.g. lib:measure:-DKNZ_MEASURE
#   with:NAME         following steps use kanzi-go_amd/variants/libknz_NAME.so (KNZ_GPU_LIB)
#   curve[:CONFIG]    single-GPU saturation curve: quick bench with --copies 1 2 4 8 16 -> <TAG>_saturation_<config>.json
#   env:K=V           export K=V for the following steps
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${TAG:-r04}
line() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print(d["config"]["workload"][:70], "| value", d["value"], "enc", d["encode_MBps"], "dec", d["decode_MBps"], "exact", d.get("bit_exact_vs_oracle"))
print("  stages", r.get("all_stage_ms")); print("  phases", r.get("phase_ms_per_step")); print("  kernels", r.get("kernel_ms_per_step"))
print("  dominant", r.get("kernel"), "frac", r.get("frac"), "traffic/alg", r.get("traffic_over_algorithmic"), "cpu", d.get("cpu_baseline", {}).get("encode_MBps"), d.get("cpu_baseline", {}).get("decode_MBps"))
PY
}
for step in "$@"; do
  what=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}
  case $what in
    env) export "$arg";;
    with) export KNZ_GPU_LIB=$PWD/kanzi-go_amd/variants/libknz_$arg.so;;
    lib) name=${arg%%:*}; flags=${arg#*:}; mkdir -p kanzi-go_amd/variants
         /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-variable -Wno-unused-value $flags -o kanzi-go_amd/variants/libknz_$name.so kanzi-go_amd/csrc/knz_gpu.hip 2>&1 | grep -i "error"; echo "lib $name done";;
    tests) timeout 2400 python -m pytest tests -m gpu -x -q ${arg:+-k "$arg"} --durations=6 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -10 gpurun_out/${TAG}_pytest.log;;
    bench) c=${arg:-bwt}; extra=""; [ $c = fpaq ] && extra="--steps 1 --warmup 0 --no-host-hook"
         timeout 3000 python bench.py --config $c $extra > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err; echo "bench $c rc=$?"; line gpurun_out/${TAG}_bench_$c.json; tail -2 gpurun_out/${TAG}_bench_$c.err;;
    quick) c=${arg:-bwt}; timeout 900 python bench.py --config $c --no-cpu-baseline --no-pmc --no-host-hook --steps 5 --warmup 2 > gpurun_out/${TAG}_quick_$c.json 2> gpurun_out/${TAG}_quick_$c.err; echo "quick $c rc=$?"; line gpurun_out/${TAG}_quick_$c.json
         grep "suffix sort" gpurun_out/${TAG}_quick_$c.err | head -8;;
    trace) c=${arg:-bwt}; rm -rf gpurun_out/prof_kt
         timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --config $c --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 2 --warmup 1 > gpurun_out/${TAG}_trace_$c.log 2>&1
         db=$(find gpurun_out/prof_kt -name '*.db' | head -1)
         python tools/rocpd_summary.py $db gpurun_out/${TAG}_kernel_stats_$c.md > /dev/null; echo "trace $c rc=$?"; sed -n 3,24p gpurun_out/${TAG}_kernel_stats_$c.md
         python tools/trace_rounds.py $db > gpurun_out/${TAG}_bwt_rounds_$c.txt 2>/dev/null && cat gpurun_out/${TAG}_bwt_rounds_$c.txt
         find gpurun_out -name '*.db' -size +8M -delete;;
    curve) c=${arg:-bwt}; : > gpurun_out/${TAG}_saturation_$c.jsonl
         for k in 1 2 4 8 16; do
           timeout 1200 python bench.py --config $c --copies $k --no-cpu-baseline --no-pmc --no-host-hook --no-verify --steps 2 --warmup 1 >> gpurun_out/${TAG}_saturation_$c.jsonl 2> gpurun_out/${TAG}_saturation_$c.err || echo "copies $k failed: $(tail -1 gpurun_out/${TAG}_saturation_$c.err)"
         done
         python - gpurun_out/${TAG}_saturation_$c.jsonl gpurun_out/${TAG}_saturation_$c.json <<'PY'
import json, sys
rows = []
for l in open(sys.argv[1]):
    l = l.strip()
    if not l.startswith("{"):
        continue
    d = json.loads(l)
    rows.append({"blocks": d["config"]["blocks"], "bytes": d["config"]["blocks"] and int(d["value"] * d["ms_per_step"] * 1e3), "round_trip_MBps": d["value"], "encode_MBps": d["encode_MBps"],
                 "decode_MBps": d["decode_MBps"], "ms_per_step": d["ms_per_step"], "stage_ms": d["roofline"].get("all_stage_ms")})
json.dump({"what": "one MI355X, one stream of K copies of the corpus (bench.py --copies K): throughput against blocks in flight", "rows": rows}, open(sys.argv[2], "w"), indent=1)
for r in rows:
    print(r["blocks"], "blocks:", "round trip", r["round_trip_MBps"], "enc", r["encode_MBps"], "dec", r["decode_MBps"], "ms", r["ms_per_step"])
PY
         ;;
    *) echo "unknown step $step";;
  esac
done
