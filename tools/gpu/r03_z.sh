#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "rank or ans1" > gpurun_out/zz_tests.txt 2>&1; echo tests rc=$?; tail -3 gpurun_out/zz_tests.txt
