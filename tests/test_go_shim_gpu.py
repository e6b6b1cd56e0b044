"""The Go host driving the device, executed: the cgo shim of go/ (gpu_batch.go, gpu_stream.go, gpu_transform.go, gpu_entropy.go) and the two-line
patch of INTEGRATION.md in the reference's CompressedStream.go, translated to C++ by tools/go2cpp together with the rest of the reference and linked
against libknz_gpu.so (`make -C oracle _ref_gpu`, built in the build container; the library travels to the GPU box). The reference's own Writer /
Reader then hand their block batches to the GPU batch scheduler through the shim's Go code, and the stream they write must be the stream they write
WITHOUT the device (oracle/_ref, the same translation without shim and patch) and the stream the committed reference vectors describe."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import parity_cases as P
import ref_lib as R

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "_ref", "libknz_ref_gpu.so")


@pytest.fixture(scope="module")
def G():
    if not os.path.exists(SO):
        pytest.skip("oracle/_ref/libknz_ref_gpu.so is not built (make -C oracle _ref_gpu in the build container)")
    # torch first: it carries its own copy of the HIP runtime, and whichever copy comes up first owns the device for the process (the rest of the
    # GPU suite holds its device memory in torch tensors)
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.zeros(1, device="cuda:0")
    L = C.CDLL(SO)
    u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    L.kref_last_error.restype = C.c_char_p
    L.kref_gpu_compress.argtypes = [u8p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, C.c_int, u8p, C.c_uint64, u64p]
    L.kref_gpu_decompress.argtypes = [u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint64, u64p]
    L.kref_gpu_transform.argtypes = [C.c_int, C.c_uint64, u8p, C.c_uint64, u8p, C.c_uint64, u64p]
    L.kref_gpu_entropy_encode.argtypes = [C.c_uint32, u8p, C.c_uint64, u8p, C.c_uint64, u64p]
    return L


def _buf(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def gpu_compress(L, data, transform, entropy, bs, ck=0, jobs=16, skip=False):
    a, p = _buf(data)
    cap = len(data) + len(data) // 2 + (1 << 20)
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64()
    rc = L.kref_gpu_compress(p, len(data), transform.encode(), entropy.encode(), bs, ck, jobs, len(data), 1 if skip else 0,
                             out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n))
    assert rc == 0, (rc, L.kref_last_error())
    return out[: n.value].tobytes()


def gpu_decompress(L, stream, cap, jobs=16, may_fail=False):
    a, p = _buf(stream)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_uint64()
    rc = L.kref_gpu_decompress(p, len(stream), jobs, out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n))
    if may_fail and rc != 0:
        return None
    assert rc == 0, (rc, L.kref_last_error())
    return out[: n.value].tobytes()


def ref_decompress_or_none(stream, cap, jobs=1):
    try:
        return R.decompress(stream, cap, jobs=jobs)
    except Exception:   # noqa: BLE001 (the reference's Reader returned an error)
        return None


def check_lanes(L, cfgs, sizes, lanes_list, jobs_depth):
    """Writer / Reader.EnableGPUDevices (go/gpu_stream.go, knz_open_devices): the batches fan out over K lanes (K logical devices on ordinal 0). The
    stream the reference's Writer writes is the stream it writes without any device, whatever K, `jobs` and the batch depth; its Reader reads it back."""
    cases = 0
    for transform, entropy, bs, ck in cfgs:
        for n in sizes(bs):
            data = P.corpus(n, seed=n % 89)
            want = R.compress(data, transform, entropy, bs, ck) if R.available() else O.compress(data, transform, entropy, bs, ck)
            try:
                for lanes in lanes_list:
                    L.kref_gpu_lanes(lanes)
                    for jobs, depth in jobs_depth:
                        L.kref_gpu_depth(depth)
                        assert gpu_compress(L, data, transform, entropy, bs, ck, jobs=jobs) == want, (transform, entropy, n, lanes, jobs, depth, "stream")
                        assert gpu_decompress(L, want, n + 64, jobs=jobs) == data, (transform, entropy, n, lanes, jobs, depth, "decode")
                        cases += 1
            finally:
                L.kref_gpu_lanes(0)
                L.kref_gpu_depth(0)
    return cases


def check_ranges_and_damage(L, cfgs, lanes_list=(0,)):
    """ctx["from"] / ctx["to"] on the device path (io/CompressedStream.go:1854-1867: blocks outside the range are read and dropped) and streams that end too
    early or carry damaged framing: the Reader over the device returns what the reference's Reader returns, an error where it returns an error, and
    the process lives (the shared bitstream's panics are recovered in processBlockGPU as in decodingTask.decode, :1778-1786)."""
    cases = 0
    for transform, entropy, bs, ck in cfgs:
        n = 6 * bs + 77
        data = P.corpus(n, seed=5)
        stream = R.compress(data, transform, entropy, bs, ck)
        for lanes in lanes_list:
            L.kref_gpu_lanes(lanes)
            try:
                for frm, to in ((2, 4), (1, 2), (3, 100), (5, 5), (7, 8), (-1, 3), (4, -1)):
                    R.lib().kref_decode_range(frm, to)
                    L.kref_decode_range(frm, to)
                    for jobs in (1, 2, 16):
                        want = R.decompress(stream, n + 64, jobs=jobs)
                        lo, hi = max(frm, 1), (to if to >= 0 else 100)
                        assert want == b"".join(data[(b - 1) * bs: b * bs] for b in range(lo, min(hi, 8))), (frm, to, jobs, "reference itself")
                        assert gpu_decompress(L, stream, n + 64, jobs=jobs) == want, (transform, entropy, frm, to, jobs, lanes)
                        cases += 1
            finally:
                R.lib().kref_decode_range(-1, -1)
                L.kref_decode_range(-1, -1)
            for cut in (len(stream) - 1, len(stream) - 40, len(stream) // 2, 30, 24):
                bad = stream[:cut]
                for jobs in (1, 16):
                    want = ref_decompress_or_none(bad, n + 64, jobs=jobs)
                    got = gpu_decompress(L, bad, n + 64, jobs=jobs, may_fail=True)
                    assert (got is None) == (want is None), (transform, entropy, cut, jobs, lanes, "error or not", got is None, want is None)
                    cases += 1
            L.kref_gpu_lanes(0)
    return cases


@pytest.mark.parametrize("cfg", [("NONE", "HUFFMAN", 1 << 16, 0), ("BWT+RANK+ZRLT", "ANS1", 1 << 16, 64), ("LZ", "ANS0", 1 << 16, 32),
                                 ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0", 1 << 17, 0), ("BWT+RANK+ZRLT", "FPAQ", 1 << 16, 0), ("NONE", "NONE", 1 << 16, 0)])
def test_reference_writer_and_reader_through_the_go_shim(G, cfg):
    transform, entropy, bs, ck = cfg
    for n in (0, 1, 15, 16, 70000, 5 * bs + 4321, 40 * bs + 17):           # (40 blocks with jobs 16: three device batches per stream)
        data = P.corpus(n, seed=n % 97)
        via_gpu = gpu_compress(G, data, transform, entropy, bs, ck)
        host_only = R.compress(data, transform, entropy, bs, ck) if R.available() else O.compress(data, transform, entropy, bs, ck)
        assert via_gpu == host_only, (cfg, n, "the reference Writer writes another stream when its batches go through the device")
        assert gpu_decompress(G, host_only, n + 64) == data, (cfg, n, "reference Reader over the device")
        for jobs in (1, 3):
            assert gpu_compress(G, data, transform, entropy, bs, ck, jobs=jobs) == host_only, (cfg, n, jobs)
            assert gpu_decompress(G, via_gpu, n + 64, jobs=jobs) == data


def test_go_shim_against_the_committed_reference_vectors(G):
    import test_ref_streams as T
    data = T._inputs()
    cases = [(c, s) for c, s in T._cases() if c["block_size"] <= (4 << 20) and not c["name"].startswith("cfg5")]
    assert len(cases) > 200
    for c, ref in cases[::3]:
        src = data[c["input"][:-4]]
        got = gpu_compress(G, src, c["transform"], c["entropy"], c["block_size"], c["checksum"], skip=c["skip_blocks"])
        assert len(got) == c["stream_bytes"] and hashlib.sha256(got).hexdigest() == c["sha256"], c["name"]
        assert gpu_decompress(G, got, len(src) + 64) == src, c["name"]


def test_plugin_objects_of_the_go_shim(G):
    """kanzi.ByteTransform (go/gpu_transform.go) and kanzi.EntropyEncoder (go/gpu_entropy.go) objects over a device handle == the oracle's objects"""
    u8p = C.POINTER(C.c_uint8)
    for tname in ("RANK", "ZRLT", "LZ", "BWT", "SRT"):
        tid = P._TID[tname]
        for name, data in list(P.transform_inputs(max_len=1 << 16))[:14]:
            a, p = _buf(data)
            cap = 2 * len(data) + 65536
            out = np.zeros(cap, dtype=np.uint8)
            n = C.c_uint64()
            rc = G.kref_gpu_transform(0, tid, p, len(data), out.ctypes.data_as(u8p), cap, C.byref(n))
            o = O.transform_forward(tid, data)
            assert (rc == -1) == (o is None), (tname, name, rc, G.kref_last_error())
            if o is None:
                continue
            assert rc == 0 and out[: n.value].tobytes() == o, (tname, name)
            back = np.zeros(len(data) + 1024, dtype=np.uint8)
            fa, fp = _buf(o)
            rc = G.kref_gpu_transform(1, tid, fp, len(o), back.ctypes.data_as(u8p), len(back), C.byref(n))
            assert rc == 0 and back[: n.value].tobytes() == data, (tname, name, "inverse")
    for ename in ("HUFFMAN", "ANS0", "ANS1", "FPAQ"):
        et = O.entropy_type(ename)
        for name, data in list(P.entropy_inputs())[:16]:
            if ename == "ANS1" and len(data) in (2, 3):
                continue
            a, p = _buf(data)
            cap = 2 * len(data) + 262144
            out = np.zeros(cap, dtype=np.uint8)
            bits = C.c_uint64()
            rc = G.kref_gpu_entropy_encode(et, p, len(data), out.ctypes.data_as(u8p), cap, C.byref(bits))
            assert rc == 0, (ename, name, G.kref_last_error())
            ob, obits = O.entropy_encode(et, data)
            assert bits.value == obits and out[: (obits + 7) // 8].tobytes() == ob, (ename, name)


@pytest.mark.parametrize("cfg", [("BWT+RANK+ZRLT", "ANS1", 1 << 16, 64), ("LZ", "HUFFMAN", 1 << 15, 32), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0", 1 << 17, 0)])
def test_listeners_hear_the_same_events_through_the_go_shim(G, cfg):
    """kanzi.Listener (Definitions.go, Event.go): a Writer / Reader whose batches go to the device tells its listeners what a one-job Writer / Reader tells
    them - the same events in the same order with the same block ids, sizes (block, post-transform, compressed) and hashes, EVT_BLOCK_INFO with its
    stream offset and skip flags included (verbosity 5)."""
    if not R.available():
        pytest.skip("oracle/_ref is not here to say what the reference's listeners hear")
    transform, entropy, bs, ck = cfg
    for n in (1, 40, 3 * bs + 77, 21 * bs + 5):
        data = P.corpus(n, seed=7 + n % 13)
        R.record_events(5)
        stream = R.compress(data, transform, entropy, bs, ck, jobs=1)
        want_w = R.event_log()
        assert R.decompress(stream, n + 64, jobs=1) == data
        want_r = R.event_log()
        R.record_events(-1)
        assert len(want_w) >= 5 * max(1, -(-n // bs)) and len(want_r) >= 5 * max(1, -(-n // bs)), (cfg, n, want_w[:3])
        for jobs in (1, 16):
            R.record_events(5, G)
            assert gpu_compress(G, data, transform, entropy, bs, ck, jobs=jobs) == stream
            got_w = R.event_log(G)
            assert gpu_decompress(G, stream, n + 64, jobs=jobs) == data
            got_r = R.event_log(G)
            R.record_events(-1, G)
            assert got_w == want_w, (cfg, n, jobs, "Writer events", [(a, b) for a, b in zip(got_w, want_w) if a != b][:3], len(got_w), len(want_w))
            assert got_r == want_r, (cfg, n, jobs, "Reader events", [(a, b) for a, b in zip(got_r, want_r) if a != b][:3], len(got_r), len(want_r))


def test_batch_depth_of_its_own_through_the_go_shim(G):
    """Writer / Reader.EnableGPUDepth (go/gpu_stream.go): more blocks per device batch than `jobs` allows (the reference caps jobs at 64). 300 blocks of 16 KiB with
    depths 1, 100 and 1000: the streams are the stream of the reference without the device, and they decode at every depth."""
    bs, n = 1 << 14, 300 * (1 << 14) + 123
    data = P.corpus(n, seed=31)
    for transform, entropy, ck in (("BWT+RANK+ZRLT", "ANS1", 32), ("LZ", "HUFFMAN", 0)):
        want = R.compress(data, transform, entropy, bs, ck) if R.available() else O.compress(data, transform, entropy, bs, ck)
        try:
            for depth in (1, 100, 1000):
                G.kref_gpu_depth(depth)
                assert gpu_compress(G, data, transform, entropy, bs, ck, jobs=4) == want, (transform, entropy, depth)
                assert gpu_decompress(G, want, n + 64, jobs=4) == data, (transform, entropy, depth)
        finally:
            G.kref_gpu_depth(0)


def test_several_devices_through_the_go_shim(G):
    """Row e' of the round-5 verdict: the GPUs behind the boundary the Go host has. K = 1, 2, 3, 8 logical devices on the one GPU of the box, `jobs` 16 and a
    batch depth of 128, six pipelines: stream == reference stream, every time."""
    cfgs = [("NONE", "HUFFMAN", 1 << 16, 0), ("BWT+RANK+ZRLT", "ANS1", 1 << 16, 64), ("LZ", "ANS0", 1 << 16, 32), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0", 1 << 17, 0),
            ("BWT+RANK+ZRLT", "FPAQ", 1 << 16, 0), ("NONE", "NONE", 1 << 16, 0)]
    n = check_lanes(G, cfgs, lambda bs: (1, 70000, 5 * bs + 4321, 40 * bs + 17, 200 * (1 << 14) + 3), (1, 2, 3, 8), ((16, 0), (3, 128)))
    assert n == 6 * 5 * 4 * 2


def test_block_ranges_and_damaged_streams_through_the_go_shim(G):
    if not R.available():
        pytest.skip("oracle/_ref is not here to say what the reference's Reader does")
    check_ranges_and_damage(G, [("BWT+RANK+ZRLT", "ANS1", 1 << 15, 32), ("LZ", "HUFFMAN", 1 << 15, 0)], lanes_list=(0, 3))
