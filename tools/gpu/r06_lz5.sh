# round 6, after emit_write copies literals a thread per byte, the mark kernel walks a thread per position and the list of moved words holds a whole map: LZ tests, the fuzz, the bench line, the trace
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_lz5; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "lz or full_size or fuzz or hardware_order or stream_bit" > $O/pytest_lz.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_lz.log
timeout 400 python tools/gpu/lz_seg_fuzz.py 240 300 > $O/lz_seg_fuzz.log 2>&1; tail -1 $O/lz_seg_fuzz.log
timeout 300 python tools/gpu/lz_order_check.py 300 > $O/lz_order_check.log 2>&1; tail -1 $O/lz_order_check.log
timeout 600 python bench.py --config lz --no-pmc --no-cpu-baseline --no-host-hook --steps 5 --warmup 2 > $O/quick_lz.json 2> $O/quick.err; echo "lz rc=$?"
rocprofv3 --kernel-trace --stats -d $O/prof -- python bench.py --config lz --steps 3 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/prof_lz.json 2> $O/prof_lz.err
DB=$(find $O/prof -name "*.db" | head -1); python tools/rocpd_summary.py $DB $O/config3_lz_kernel_stats.md > /dev/null 2>&1; rm -rf $O/prof
python - $O/quick_lz.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
print("lz value", d["value"], "enc", d["encode_MBps"], "dec", d["decode_MBps"], "exact", d.get("bit_exact_vs_oracle"), d.get("bit_exact_vs_reference"), r["all_stage_ms"])
PY
sed -n 1,22p $O/config3_lz_kernel_stats.md | cut -c1-130
