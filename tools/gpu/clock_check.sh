# does the engine clock hold while the (low-occupancy) chains of the decode run? what does the performance level change?
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r05_clocks; mkdir -p $O
rocm-smi --showperflevel --showclocks > $O/before.txt 2>&1
( for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/clocks_during_auto.txt &
python bench.py --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/bench_auto.json 2> $O/bench_auto.err
wait
rocm-smi --setperflevel high > $O/set_high.txt 2>&1
rocm-smi --showperflevel --showclocks >> $O/set_high.txt 2>&1
( for i in $(seq 1 40); do rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/clocks_during_high.txt &
python bench.py --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-host-hook --no-verify > $O/bench_high.json 2> $O/bench_high.err
wait
rocm-smi --setperflevel auto >> $O/set_high.txt 2>&1
python - <<'PY'
import json
for k in ("auto", "high"):
    d = json.loads(open(f"gpurun_out/r05_clocks/bench_{k}.json").read().strip().splitlines()[-1])
    print(k, d["value"], d["ms_per_step"], d["roofline"].get("encode_ms"), d["roofline"].get("decode_ms"))
PY
head -12 $O/before.txt; sort $O/clocks_during_auto.txt | uniq -c | sort -rn | head -5; cat $O/set_high.txt | head -12; sort $O/clocks_during_high.txt | uniq -c | sort -rn | head -5
