#!/bin/bash
mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline --steps 10 ${BENCH_ARGS} > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; echo "bench rc=$?"
python -c "
import json; d=json.loads(open('gpurun_out/bench_q.json').read().strip().splitlines()[-1]); print(d['value'], d['encode_MBps'], d['decode_MBps'], d['roofline']['all_stage_ms'], d['bit_exact_vs_oracle'])"
