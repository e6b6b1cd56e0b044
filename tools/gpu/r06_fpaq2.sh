# round 6: the FPAQ cases of the GPU suite and the chain rate. (Run on an experiment that is NOT in the tree: both walks on the scalar unit, low / high in scalar
# registers, the 4 x 256 probabilities in sixteen vector registers read and written lane by lane (v_readlane / v_writelane through M0), no LDS on the chain.
# Bit-exact, and exactly as fast as the one-lane vector form that ships: 0.95 / 0.88 MB/s per chain; ~40 instructions per bit as compiled either way, and a
# lone wave issues one per ~3 ns whatever unit it goes to. Halving that needs the walk written by hand in ISA; dropped)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_fpaq2; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py -x -q -k "FPAQ or fpaq or entropy_objects or stream_bit or fuzz or reference_streams or ref_own" > $O/pytest_fpaq.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_fpaq.log
timeout 600 python tools/gpu/fpaq_chain_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/fpaq_chain_rate.txt
