"""The Go host driving the device, on the HIP emulator (CPU suite): oracle/_ref/libknz_ref_gpu.so (the reference + the cgo shim of go/ + the patch of
INTEGRATION.md, translated) runs in a child process with tests/emu's build of the library preloaded in place of libknz_gpu.so. The streams the
reference's Writer writes through the shim must be the streams it writes without it, its Reader must read them back, and the listeners must hear the
same events (tests/go_shim_emu_check.py). The real-device form of the same checks is tests/test_go_shim_gpu.py."""
import os
import subprocess
import sys

import pytest

import ref_lib as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_streams_and_events_through_the_go_shim_on_the_emulator():
    if not R.can_build():
        pytest.skip("/root/reference is not here: oracle/_ref_gpu cannot be (re)built against the current shim")
    import parity_cases as P
    P.EmuBackend()                                        # (builds tests/emu/build/libknz_gpu_emu.so when it is stale)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref", "_ref_gpu"])
    env = dict(os.environ, LD_PRELOAD=os.path.join(ROOT, "tests", "emu", "build", "libknz_gpu_emu.so"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "go_shim_emu_check.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=1500)
    assert r.returncode == 0 and "cases ok" in r.stdout, r.stdout[-3000:]
