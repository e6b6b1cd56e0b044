# one stream, deeper batches at the host hook (what Writer / Reader.EnableGPUDepth of go/gpu_stream.go hand over per call): PCIe-inclusive rates against blocks per batch
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r05_hook_depth; mkdir -p $O
for c in 1 2 4 5; do
  timeout 900 python bench.py --copies $c --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > $O/bwt_copies_$c.json 2> $O/bwt_copies_$c.err
done
for c in 1 2 4 8; do
  timeout 900 python bench.py --config l5 --copies $c --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > $O/l5_copies_$c.json 2> $O/l5_copies_$c.err
done
python - <<'PY'
import json, glob, os
O = "gpurun_out/r05_hook_depth"
rows = []
for f in sorted(glob.glob(O + "/*_copies_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        rows.append({"file": os.path.basename(f), "error": str(e)}); continue
    hh = d.get("host_hook_MBps") or {}
    rows.append({"file": os.path.basename(f), "blocks": d["config"].get("blocks"), "device_resident_round_trip": d["value"], "hook_encode": hh.get("encode"), "hook_decode": hh.get("decode"), "hook_round_trip": hh.get("round_trip"), "hook_ok": hh.get("ok")})
json.dump(rows, open(O + "/summary.json", "w"), indent=1)
for r in rows: print(r)
PY
