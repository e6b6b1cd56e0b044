# round 5, final state: the reference's own tests on the device, the whole GPU suite, the default bench (one gpurun call)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r05_final2; mkdir -p $O
timeout 900 python -m pytest tests/test_ref_own_tests.py tests/test_go_shim_gpu.py -m gpu -q -x 2>&1 | tail -25 > $O/own_tests.log
timeout 600 ./oracle/_ref/knz_ref_gpu_tests io.TestCompressedStream entropy.TestHuffman entropy.TestANS0 entropy.TestANS1 entropy.TestFPAQ entropy.TestFPAQCodecSpecificPatterns transform.TestLZ transform.TestLZX transform.TestLZP transform.TestZRLT transform.TestSRT transform.TestRank transform.TestMTFT transform.TestTextCodec transform.TestUTFCodec transform.TestLZCodecSpecifics transform.TestUTFCodecMinBlockAndRoundTrip transform.TestTextCodecMinBlockAndRoundTrip > $O/own_tests_direct.log 2> $O/own_tests_direct.err; echo "exit $?" >> $O/own_tests_direct.log
tail -c 3000 $O/own_tests_direct.err > $O/own_tests_direct.err.tail; rm -f $O/own_tests_direct.err
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/gpu_suite.log
python bench.py > $O/config_bwt_bench.json 2> $O/bwt.err
tail -3 $O/own_tests.log; tail -4 $O/own_tests_direct.log; tail -3 $O/gpu_suite.log; cat $O/config_bwt_bench.json | cut -c1-400
