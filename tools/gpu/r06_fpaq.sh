cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
mkdir -p gpurun_out/r06_fpaq
timeout 600 python bench.py --config lz --no-pmc --no-cpu-baseline --no-host-hook --steps 3 --warmup 1 > gpurun_out/r06_fpaq/lz_quick.json 2> gpurun_out/r06_fpaq/lz_quick.err; cut -c1-330 gpurun_out/r06_fpaq/lz_quick.json | tail -c 250; echo
timeout 2300 python bench.py --config fpaq --steps 1 --warmup 0 --no-host-hook > gpurun_out/r06_fpaq/config_fpaq_bench.json 2> gpurun_out/r06_fpaq/fpaq.err; echo rc=$?; cut -c1-400 gpurun_out/r06_fpaq/config_fpaq_bench.json
