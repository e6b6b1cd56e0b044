cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_mid; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -14 $O/gpu_suite.log
timeout 400 python tools/gpu/ext_fuzz.py 240 6100 > $O/ext_fuzz.log 2>&1; tail -2 $O/ext_fuzz.log
timeout 900 python bench.py --config lz > $O/config_lz_bench.json 2> $O/lz.err; echo "lz rc=$?"
python - $O/config_lz_bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; c = d.get("cpu_baseline", {})
print("lz value", d["value"], "enc", d["encode_MBps"], "dec", d["decode_MBps"], "exact", d.get("bit_exact_vs_oracle"), d.get("bit_exact_vs_reference"))
print(" dominant", r.get("kernel"), "frac", r.get("frac"), "traffic/alg", r.get("traffic_over_algorithmic"), "| cpu", c.get("kind"), c.get("encode_MBps"), c.get("decode_MBps"), "cores", c.get("cores"), "gpu/cpu enc", c.get("gpu_encode_over_cpu_encode"))
print(" hook", d.get("host_hook_MBps"))
PY
