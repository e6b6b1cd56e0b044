#!/bin/bash
# round-end evidence: full default bench line (with cpu_baseline), kernel trace + PMC traffic of the same command, the other
# configs' bench lines, the host-pointer hook rate
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo "bench rc=$?"
rm -rf gpurun_out/pmc_f gpurun_out/pmc_w gpurun_out/prof_kt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_f -o f -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_w -o w -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > gpurun_out/pmc_w.log 2>&1
python tools/pmc_traffic.py $(find gpurun_out/pmc_f -name '*.db' | head -1) $(find gpurun_out/pmc_w -name '*.db' | head -1) gpurun_out/pmc_traffic.json > /dev/null; echo "pmc rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt -o kt -- python bench.py --no-cpu-baseline --steps 4 --warmup 1 > gpurun_out/prof_kt.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt -name '*.db' | head -1) gpurun_out/kernel_stats_config2.md > /dev/null
timeout 300 python tools/host_hook_rate.py > gpurun_out/host_hook.json 2> gpurun_out/host_hook.err; echo "hook rc=$?"; cat gpurun_out/host_hook.json
for c in ans0 lz bwt; do
  timeout 600 python bench.py --config $c --steps 2 --warmup 1 > gpurun_out/bench_$c.json 2> gpurun_out/bench_$c.err; echo "$c rc=$?"
done
rm -rf gpurun_out/prof_kt4
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_kt4 -o kt -- python bench.py --config bwt --no-cpu-baseline --no-verify --steps 2 --warmup 1 > gpurun_out/prof_kt4.log 2>&1
python tools/rocpd_summary.py $(find gpurun_out/prof_kt4 -name '*.db' | head -1) gpurun_out/kernel_stats_config4.md > /dev/null
find gpurun_out -name '*.db' -size +8M -delete
python - <<'PY'
import json
for n in ['full','ans0','lz','bwt']:
    try:
        d=json.loads(open(f'gpurun_out/bench_{n}.json').read().strip().splitlines()[-1])
        print(n, d['value'], d['encode_MBps'], d['decode_MBps'], d['roofline']['kernel'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('encode_MBps'), d.get('cpu_baseline',{}).get('decode_MBps'), d['bit_exact_vs_oracle'])
    except Exception as e: print(n,'ERR',e)
PY
