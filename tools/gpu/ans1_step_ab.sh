# A/B of the order-1 rANS decoder's hand-written step: one LDS round trip per step (DPP hand-over, default) against round 4's two (KNZ_ANS1_LOHI_LDS)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r05_ans1_step; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -q -x -k "ans1 or rank_pipe or entropy_objects or device_vs_ref or full_size_config4" 2>&1 | tail -4 > $O/tests.log
python bench.py --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > $O/bench_new.json 2> $O/bench_new.err
KNZ_ANS1_LOHI_LDS=1 python bench.py --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook > $O/bench_old.json 2> $O/bench_old.err
python - <<'PY'
import json
for k in ("new", "old"):
    try:
        d = json.loads(open(f"gpurun_out/r05_ans1_step/bench_{k}.json").read().strip().splitlines()[-1])
        km = d.get("kernel_ms") or d["roofline"].get("kernel_ms") or {}
        print(k, d["value"], d["ms_per_step"], d["roofline"].get("encode_ms"), d["roofline"].get("decode_ms"), d.get("bit_exact_vs_oracle"), {a: b for a, b in (d.get("all_stage_ms") or {}).items()} if isinstance(d.get("all_stage_ms"), dict) else "")
    except Exception as e:
        print(k, "error", e)
PY
tail -3 $O/tests.log
