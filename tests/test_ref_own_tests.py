"""The reference's OWN unit tests (kanzi-go's *_test.go files), translated by tools/go2cpp with the sources they test — TEST INFRASTRUCTURE.

1. On the translation itself (`make -C oracle _ref_tests` -> oracle/_ref/knz_ref_tests): every `func TestXxx` of entropy/Entropy_test.go,
   transform/Transforms_test.go, transform/BWT_test.go, transform/EXECodec_test.go, bitstream/DefaultBitstream_test.go and io/CompressedStream_test.go
   must pass, as `go test ./...` would have it. This is the check of oracle/_ref that does not go through anything written here.
2. On the device (`make -C oracle _ref_gpu_tests` -> oracle/_ref/knz_ref_gpu_tests): the same test files, with the three factory helpers they build their
   objects through (getEncoder / getDecoder, getTransform, compress's Writer / Reader) re-pointed by insertions (tools/go2cpp/apply_test_patch.py) at the
   device objects of go/ (hooks: go/testhooks). The tests themselves are the reference's, unchanged: inputs, calls and checks.
   CPU suite: entropy and transform tests against the kernels on the HIP emulator (the library under tests/emu preloaded in place of libknz_gpu.so);
   GPU suite: all of them, streams included, on the MI355X, and the count of device-backed objects the tests were handed must not be zero.
"""
import os
import subprocess

import pytest

import ref_lib as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE = os.path.join(ROOT, "oracle")
CPU_BIN = os.path.join(ORACLE, "_ref", "knz_ref_tests")
GPU_BIN = os.path.join(ORACLE, "_ref", "knz_ref_gpu_tests")

# what the device implements of the reference's test list (the other tests of the same files run on the reference's own objects and prove nothing about the device)
DEVICE_ENTROPY = ["entropy.TestHuffman", "entropy.TestANS0", "entropy.TestANS1", "entropy.TestFPAQ", "entropy.TestFPAQCodecSpecificPatterns"]
DEVICE_TRANSFORM = ["transform.TestLZ", "transform.TestLZX", "transform.TestLZP", "transform.TestZRLT", "transform.TestSRT", "transform.TestRank", "transform.TestMTFT",
                    "transform.TestTextCodec", "transform.TestUTFCodec", "transform.TestLZCodecSpecifics", "transform.TestUTFCodecMinBlockAndRoundTrip",
                    "transform.TestTextCodecMinBlockAndRoundTrip"]
DEVICE_IO = ["io.TestCompressedStream"]


def _binary(path, target):
    if R.can_build():
        subprocess.check_call(["make", "-s", "-C", ORACLE, target])
    if not os.path.exists(path):
        pytest.skip(f"{os.path.relpath(path, ROOT)} is not built and /root/reference is not here to build it from")
    return path


def _run(path, names, env=None, timeout=1500):
    e = dict(os.environ)
    e.update(env or {})
    e.setdefault("KREF_TEST_SEED", "20260926")
    r = subprocess.run([path] + list(names), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=e, timeout=timeout)
    lines = r.stdout.strip().splitlines()
    results = {l.split()[0]: l.split()[1] for l in lines if len(l.split()) == 2 and l.split()[1] in ("ok", "FAIL")}
    return r.returncode, results, r.stdout


def test_reference_unit_tests_pass_on_the_translation():
    path = _binary(CPU_BIN, "_ref_tests")
    names = subprocess.run([path, "--list"], stdout=subprocess.PIPE, text=True, check=True).stdout.split()
    assert len(names) >= 52 and {"entropy.TestANS1", "entropy.TestCM", "transform.TestBWT", "transform.TestLZCodecSpecifics", "io.TestCompressedStream"} <= set(names)
    rc, results, out = _run(path, [])
    failed = [n for n in names if results.get(n) != "ok"]
    assert rc == 0 and not failed, out[-4000:]


def _device_objects(out):
    line = [l for l in out.splitlines() if l.startswith("device objects:")][-1].split()
    return int(line[3]), int(line[5]), int(line[7])


def test_reference_unit_tests_on_the_emulated_device():
    import parity_cases as P
    be = P.EmuBackend()                                   # (builds tests/emu/build/libknz_gpu_emu.so from the kernel sources when it is stale)
    path = _binary(GPU_BIN, "_ref_gpu_tests")
    emu = os.path.join(ROOT, "tests", "emu", "build", "libknz_gpu_emu.so")
    assert os.path.exists(emu)
    rc, results, out = _run(path, DEVICE_ENTROPY + DEVICE_TRANSFORM, env={"LD_PRELOAD": emu})
    assert rc == 0 and all(results.get(n) == "ok" for n in DEVICE_ENTROPY + DEVICE_TRANSFORM), out[-4000:]
    ent, trf, _ = _device_objects(out)
    assert ent >= 100 and trf >= 300, out[-300:]          # the tests really were handed device objects


@pytest.mark.gpu
def test_reference_unit_tests_on_the_device():
    path = _binary(GPU_BIN, "_ref_gpu_tests")
    names = DEVICE_ENTROPY + DEVICE_TRANSFORM + DEVICE_IO
    rc, results, out = _run(path, names)
    assert rc == 0 and all(results.get(n) == "ok" for n in names), out[-4000:]
    ent, trf, streams = _device_objects(out)
    assert ent >= 100 and trf >= 300 and streams >= 20, out[-300:]
