# round 6, after the register windows of the lane-per-segment LZ parse became shifts and cand[] moved to LDS lines: tests, stress, the bench line, the trace
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_lzf; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "lz or full_size or fuzz or hardware_order or stream_bit" > $O/pytest_lz.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_lz.log
timeout 400 python tools/gpu/lz_seg_fuzz.py 240 300 > $O/lz_seg_fuzz.log 2>&1; tail -1 $O/lz_seg_fuzz.log
timeout 300 python tools/gpu/lz_order_check.py 300 > $O/lz_order_check.log 2>&1; tail -1 $O/lz_order_check.log
timeout 900 python bench.py --config lz > $O/config_lz_bench.json 2> $O/lz.err; echo "lz rc=$?"
python tools/gpu/lz_rounds.py 2>&1 | grep -v amdgpu.ids | grep -i "rounds\|parse" > $O/lz_rounds.txt; cat $O/lz_rounds.txt
rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_lz.json 2> $O/prof_lz.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py $DB $O/config3_lz_kernel_stats.md > /dev/null 2>&1; rm -rf $O/prof
python - $O/config_lz_bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; c = d.get("cpu_baseline", {})
print("lz value", d["value"], "enc", d["encode_MBps"], "dec", d["decode_MBps"], "exact", d.get("bit_exact_vs_oracle"), d.get("bit_exact_vs_reference"), r["all_stage_ms"])
print(" dominant", r.get("kernel"), "frac", r.get("frac"), "traffic/alg", r.get("traffic_over_algorithmic"), "| cpu", c.get("kind"), c.get("encode_MBps"), c.get("decode_MBps"), "cores", c.get("cores"), "gpu/cpu enc", c.get("gpu_encode_over_cpu_encode"))
print(" hook", d.get("host_hook_MBps"))
PY
sed -n 1,14p $O/config3_lz_kernel_stats.md | cut -c1-130
