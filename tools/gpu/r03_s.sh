#!/bin/bash
mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
KNZ_BWT_PROF=1 timeout 400 rocprofv3 --kernel-trace -d gpurun_out/prof_s -o kt --output-format csv -- python bench.py --no-cpu-baseline --no-verify --no-pmc --no-host-hook --steps 1 --warmup 1 > gpurun_out/s_prof.log 2>&1; echo rc=$?
f=$(find gpurun_out/prof_s -name "*kernel_trace.csv" | head -1); echo $f
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last encode = find last occurrence of knz_bwt_init_kernel
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('knz_bwt_init_kernel')]
start = idx[-1]
end = next(i for i in range(start, len(rows)) if rows[i]['Kernel_Name'].startswith('knz_ans1_encode_kernel'))
t0 = int(rows[start]['Start_Timestamp'])
out = open('gpurun_out/s_encode_trace.txt', 'w')
for r in rows[start:end + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.write("%9.3f %8.3f %s grid=%s\n" % ((s - t0) / 1e6, (e - s) / 1e6, r['Kernel_Name'][:60], r.get('Grid_Size', '')))
out.close()
print("launches", end - start + 1, "span ms", (int(rows[end]['End_Timestamp']) - t0) / 1e6)
PY
grep -i "unresolved\|round" gpurun_out/s_prof.log | tail -30
rm -rf gpurun_out/prof_s
