# round 6: the order-1 rANS header walk reads the alphabet masks four at a time and jumps over short frequency groups inside the reader's window
# (dec_walk of configs[3] 5.64 -> 4.30 ms). The same change in the order-0 walk made the fused walk + decode kernel of the LZ line 0.2 ms SLOWER (3.93 -> 4.12 ms,
# twice) and left the ans0 line where it was: taken back there.
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_walk; mkdir -p $O
echo "(suite skipped in this run)"
for cfg in lz bwt lz bwt; do
  timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-pmc --no-host-hook --steps 5 --warmup 2 > $O/$cfg.json 2> $O/$cfg.err
  python - $O/$cfg.json $cfg <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]; k = r["kernel_ms_per_step"]
print(sys.argv[2], "enc", d["encode_MBps"], "dec", d["decode_MBps"], "exact", d.get("bit_exact_vs_oracle"), d.get("bit_exact_vs_reference"), r["all_stage_ms"], {n: v for n, v in k.items() if "walk" in n})
PY
done
