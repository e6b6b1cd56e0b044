# round 6, first GPU trip: the new multi-lane scheduler (tests, go shim, deep-batch stress), the default bench line, and the in-process-devices curves
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_first; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "multi_device or deep_batches or go_shim or rank_pipe or ans1 or block_batch or concurrent_handles or split_when" --durations=8 > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -14 $O/pytest_new.log
timeout 900 python bench.py --no-pmc --steps 5 --warmup 2 > $O/bench_bwt.json 2> $O/bench_bwt.err; echo "bench rc=$?"; cut -c1-400 $O/bench_bwt.json
for c in bwt lz huffman; do
  timeout 900 python bench.py --config $c --in-process-devices 1,2,3,4,8 > $O/lanes_$c.json 2> $O/lanes_$c.err; echo "lanes $c rc=$?"
  python - $O/lanes_$c.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for r in d["curve"]:
    print(r["lanes"], "lanes:", "enc", r["encode_MBps"], "dec", r["decode_MBps"], "rt", r["round_trip_MBps"], "ok", r["ok"], "lane ms enc", r["lane_ms_encode"], "dec", r["lane_ms_decode"])
PY
done
timeout 900 python bench.py --config bwt --in-process-devices 1,2,4,8 --depth 104 > $O/lanes_bwt_depth104.json 2> $O/lanes_bwt_depth104.err; echo "lanes bwt depth 104 rc=$?"; cut -c1-1500 $O/lanes_bwt_depth104.json
