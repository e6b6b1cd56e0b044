"""Parity cases shared by the emulator run (CPU container, `-m "not gpu"`) and the real MI355X run (`-m gpu`).
Every case compares the HIP path (through the C ABI of include/knz_gpu.h) with the CPU oracle, bit for bit."""
import numpy as np

import knz
import oracle_lib as O

K = knz.package()


class EmuBackend:
    """TEST INFRASTRUCTURE: kernels compiled against tests/emu; 'device' memory is host memory."""
    name = "emu"

    def __init__(self):
        self.lib = knz.emu_library()

    def empty(self, n, align=16):
        raw = np.zeros(n + align + 8, dtype=np.uint8)
        off = (-raw.ctypes.data) % align
        v = raw[off:off + n + 8]
        return v.ctypes.data, (raw, v)

    def to_dev(self, data, align=16):
        ptr, keep = self.empty(len(data), align)
        keep[1][: len(data)] = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        return ptr, keep

    def to_host(self, keep, n):
        return keep[1][:n].tobytes()

    def sync(self):
        pass


class GpuBackend:
    """The product path: libknz_gpu.so (gfx950) on cuda:0, device memory owned by torch."""
    name = "gpu"

    def __init__(self):
        import torch
        self.torch = torch
        assert torch.cuda.is_available(), "GPU tests need a GPU"
        self.lib = K.build_library()      # in-tree libknz_gpu.so; (re)built with hipcc only when missing or older than its sources
        self.dev = torch.device("cuda:0")

    def empty(self, n, align=16):
        t = self.torch.zeros(n + 16, dtype=self.torch.uint8, device=self.dev)  # torch allocations are 256-byte aligned
        return t.data_ptr(), t

    def to_dev(self, data, align=16):
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data
        t = self.torch.zeros(len(a) + 16, dtype=self.torch.uint8, device=self.dev)
        if len(a):
            t[: len(a)] = self.torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        return t.data_ptr(), t

    def to_host(self, keep, n):
        return keep[:n].cpu().numpy().tobytes()

    def sync(self):
        self.torch.cuda.synchronize()


def corpus(n, seed=3):
    r = np.random.default_rng(seed)
    words = [bytes(r.integers(97, 123, int(r.integers(2, 9)), dtype=np.uint8)) for _ in range(500)]
    text = b" ".join(words[int(i)] for i in r.integers(0, 500, max(n // 4, 8)))
    binary = (r.integers(0, 256, n // 4 + 1, dtype=np.uint8) & r.integers(0, 256, n // 4 + 1, dtype=np.uint8)).tobytes()
    return (text + binary + bytes(n // 8) + text[::-1] + bytes(n))[:n]


def fib_chunk(n=16384, k=24, seed=1):
    """Fibonacci-like symbol counts: Huffman depth > 12 => exercises limitCodeLengths (HuffmanCodec.go:216-297)."""
    f = [1, 1]
    while len(f) < k:
        f.append(f[-1] + f[-2])
    f = np.array(f, dtype=np.float64)
    f = np.maximum(1, np.floor(f / f.sum() * n)).astype(np.int64)
    f[-1] += n - f.sum()
    data = np.repeat(np.arange(k, dtype=np.uint8) * 7 + 3, f)
    np.random.default_rng(seed).shuffle(data)
    return data[:n].tobytes()


def entropy_inputs():
    rng = np.random.default_rng(0x4B414E5A)
    yield "40x2", bytes([2] * 40)
    yield "ascii16", bytes([0x3d, 0x4d, 0x54, 0x47, 0x5a, 0x36, 0x39, 0x26, 0x72, 0x6f, 0x6c, 0x65, 0x3d, 0x70, 0x72, 0x65])
    yield "alt40", bytes(2 + (i & 1) for i in range(40))
    yield "one", bytes([42])
    yield "two", bytes([42, 42])
    for ii in (7, 13, 19):
        yield f"rand256_{ii}", bytes(int(64 + 4 * ii + rng.integers(0, 8 * ii + 1)) & 255 for _ in range(256))
    yield "all256", bytes(range(256))
    yield "1024x42", bytes([42] * 1024)
    yield "AB512", b"AB" * 512
    yield "rand4096", rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()
    v = bytearray(4096)
    for i in range(1, 256):
        v[i * 16] = i
    yield "sparse4096", bytes(v)
    for n in (31, 32, 33, 16383, 16384, 16385, 16384 + 31, 16384 + 32, 16384 * 3 + 5, 70001):
        yield f"geom{n}", np.minimum(rng.geometric(0.2, n), 255).astype(np.uint8).tobytes()
    yield "skew50000", (rng.integers(0, 4, 50000) * 17).astype(np.uint8).tobytes()
    yield "fib24", fib_chunk(16384, 24)
    yield "fib30", fib_chunk(16384, 30, 2) + fib_chunk(9000, 20, 3)
    yield "text", corpus(100000)


def check_entropy_encode(be, etype_name):
    c = K.Codec("NONE", etype_name, 1 << 16, lib=be.lib)
    enc = K.EntropyEncoder(c, etype_name)
    dec = K.EntropyDecoder(c, etype_name)
    et = O.entropy_type(etype_name)
    for name, data in entropy_inputs():
        gb, gbits = enc.write(data)
        ob, obits = O.entropy_encode(et, data)
        assert gbits == obits, (name, gbits, obits)
        assert gb == ob, name
        # the device decoder must accept the oracle's bits and vice versa (decoder restated independently)
        dd, used = dec.read(ob, len(data))
        assert dd == data, name
        assert used == obits, (name, used, obits)
        assert O.entropy_decode(et, gb, len(data))[0] == data
    c.close()


def check_stream(be, transform, entropy, block_size, n, seed=3):
    data = corpus(n, seed)
    c = K.Codec(transform, entropy, block_size, lib=be.lib)
    src, ksrc = be.to_dev(data)
    cap = 2 * n + 262144 * (n // block_size + 2)
    dst, kdst = be.empty(cap)
    nb = c.dev_compress(src, n, dst, cap)
    got = be.to_host(kdst, nb)
    exp = O.compress(data, transform, entropy, block_size)
    assert len(got) == len(exp), (len(got), len(exp))
    assert got == exp
    # decode the oracle's stream on the device
    sp, ks = be.to_dev(exp, 4)
    out, kout = be.empty(n + 64)
    nd = c.dev_decompress(sp, len(exp), out, n + 64)
    assert nd == n
    assert be.to_host(kout, nd) == data
    if entropy == "HUFFMAN":      # an encoder-written stream never needs the serial Huffman decoder
        assert c.last_counter(0) == 0
    c.close()
    return len(exp)


def check_block_batch(be, transform, entropy, block_size, nblocks, last_len):
    """Writer.processBlock / Reader.processBlock batch hook through host buffers (knz_encode_blocks/knz_decode_blocks)."""
    c = K.Codec(transform, entropy, block_size, lib=be.lib)
    bb = K.BlockBatch(c)
    blocks = [corpus(block_size, 10 + i) for i in range(nblocks - 1)] + [corpus(last_len, 99)]
    res = bb.encode(blocks)
    tt, et = O.transform_type(transform), O.entropy_type(entropy)
    for blk, (bits, written, mode, post, skip) in zip(blocks, res):
        o = O.encode_block(blk, tt, et)
        assert written == o["written"]
        assert bits == o["bits"]
        assert mode == o["mode"] and post == o["post_len"]
        assert skip == o["skip_flags"], (skip, o["skip_flags"])
    back = bb.decode([r[0] for r in res])
    assert back == blocks
    c.close()


def check_multi_device_batch(be, transform, entropy, block_size, nblocks, last_len, lanes, checksum_bits=0):
    """knz_open_devices: the batch hook fanned out over `lanes` lanes (io/CompressedStream.go:621-710 over GPUs instead of goroutines). The lanes are
    logical devices on ordinal 0 (one GPU box / the emulator): partition into contiguous balanced ranges, a ragged last range, lanes without blocks.
    Every block must carry the bytes, bit count, mode, post-transform length and skip flags the oracle's encodingTask.encode gives, whatever the
    number of lanes; decode through the same handle must give the blocks back."""
    c = K.Codec(transform, entropy, block_size, checksum_bits=checksum_bits, lib=be.lib, devices=[0] * lanes)
    assert c.L.knz_lane_count(c.h) == lanes
    bb = K.BlockBatch(c)
    blocks = [corpus(block_size, 10 + i) for i in range(nblocks - 1)] + [corpus(last_len, 99)]
    res = bb.encode(blocks)
    took = c.lane_times()
    q, rem = divmod(nblocks, lanes)
    assert [t[1] for t in took] == [q + (1 if l < rem else 0) for l in range(lanes)], took     # dist.block_range's rule
    assert all(t[0] == 0 for t in took)
    tt, et = O.transform_type(transform), O.entropy_type(entropy)
    O.set_ctx(block_size, et)                          # ctx["blockSize"] / ctx["entropy"] as the Writer hands them to its tasks (the TEXT codec reads both)
    for i, (blk, (bits, written, mode, post, skip)) in enumerate(zip(blocks, res)):
        o = O.encode_block(blk, tt, et, checksum_bits)
        assert written == o["written"], (transform, entropy, lanes, i, written, o["written"])
        assert bits == o["bits"], (transform, entropy, lanes, i, "bytes differ", [k for k in range(min(len(bits), len(o["bits"]))) if bits[k] != o["bits"][k]][:4])
        assert mode == o["mode"] and post == o["post_len"] and skip == o["skip_flags"]
    O.set_ctx()
    assert bb.decode([r[0] for r in res]) == blocks
    # a damaged payload in the range of a later lane: the call fails with that block's code, the blocks of the other lanes are decoded all the same
    if nblocks >= 2 and entropy != "NONE" and len(res[-1][0]) > 40:
        bad = [r[0] for r in res]
        bad[-1] = bad[-1][:1] + bytes([bad[-1][1] ^ 0xFF]) * 3 + b"\xff" * (len(bad[-1]) - 4)
        try:
            bb.decode(bad)
            failed = None
        except K.KnzError as e:
            failed = e.code
        assert failed, "a damaged block went through"
    # the single-object entry points accept the handle too (first lane)
    enc = K.EntropyEncoder(c, entropy)
    assert enc.write(blocks[0][:5000])[0] == K.EntropyEncoder(K.Codec(transform, entropy, block_size, lib=be.lib), entropy).write(blocks[0][:5000])[0]
    c.close()


def check_alloc_split(be, monkeypatch):
    """A batch whose workspace the device refuses (KNZ_TEST_ALLOC_LIMIT: bytes one workspace buffer may hold) is taken in halves by
    knz_encode_blocks / knz_decode_blocks after the handle has given its workspace back: same bytes as the unrestricted batch, and an
    error (not a crash) when even a single block does not fit."""
    for transform, entropy, limit in (("NONE", "HUFFMAN", 1200000), ("LZ", "ANS0", 1500000)):
        monkeypatch.delenv("KNZ_TEST_ALLOC_LIMIT", raising=False)
        bs, nblocks = 1 << 16, 12
        blocks = [corpus(bs, 20 + i) for i in range(nblocks - 1)] + [corpus(4321, 77)]
        c = K.Codec(transform, entropy, bs, lib=be.lib)
        want = K.BlockBatch(c).encode(blocks)
        c.close()
        monkeypatch.setenv("KNZ_TEST_ALLOC_LIMIT", str(limit))       # the stage buffers of 12 x 64 KiB blocks do not fit, those of 3-6 blocks do
        c = K.Codec(transform, entropy, bs, lib=be.lib)
        bb = K.BlockBatch(c)
        got = bb.encode(blocks)
        assert got == want, (transform, entropy, "split batch encodes differently")
        assert bb.decode([r[0] for r in got]) == blocks
        c.close()
        monkeypatch.setenv("KNZ_TEST_ALLOC_LIMIT", "1000")            # nothing fits: the call comes back with an error
        c = K.Codec(transform, entropy, bs, lib=be.lib)
        with pytest_raises_knz_any():
            K.BlockBatch(c).encode(blocks)
        c.close()
    # the same through a handle over several lanes (knz_open_devices): every lane splits and retries its own range, and a lane that cannot even take one
    # block fails the call with its error while the handle stays usable
    monkeypatch.delenv("KNZ_TEST_ALLOC_LIMIT", raising=False)
    bs, nblocks = 1 << 16, 13
    blocks = [corpus(bs, 40 + i) for i in range(nblocks - 1)] + [corpus(999, 78)]
    c = K.Codec("NONE", "HUFFMAN", bs, lib=be.lib)
    want = K.BlockBatch(c).encode(blocks)
    c.close()
    monkeypatch.setenv("KNZ_TEST_ALLOC_LIMIT", "500000")                 # (a lane's 4-5 blocks do not fit, 2-3 do)
    c = K.Codec("NONE", "HUFFMAN", bs, lib=be.lib, devices=[0, 0, 0])
    bb = K.BlockBatch(c)
    got = bb.encode(blocks)
    assert got == want, "split batches of a multi-lane handle encode differently"
    assert bb.decode([r[0] for r in got]) == blocks
    monkeypatch.setenv("KNZ_TEST_ALLOC_LIMIT", "1000")
    with pytest_raises_knz_any():
        bb.encode(blocks)
    monkeypatch.setenv("KNZ_TEST_ALLOC_LIMIT", "500000")
    assert bb.encode(blocks) == want, "the handle did not recover from a refused batch"
    c.close()
    monkeypatch.delenv("KNZ_TEST_ALLOC_LIMIT", raising=False)
    # the order-1 rANS encoder's expanded-step workspace (64 MiB per chunk slot) is halved until the device takes it: 6 slots -> 3 -> 2
    monkeypatch.setenv("KNZ_TEST_ALLOC_LIMIT", "200000000")
    check_stream(be, "NONE", "ANS1", 1 << 16, 6 * (1 << 16) - 77, seed=11)
    monkeypatch.delenv("KNZ_TEST_ALLOC_LIMIT", raising=False)


class pytest_raises_knz_any:
    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        assert et is not None and issubclass(et, K.KnzError), "the call succeeded although no workspace buffer may exist"
        return True


def check_assemble(be, entropy, block_size, n, ranks):
    """Multi-GPU sharding: each 'rank' encodes a contiguous range of blocks, rank 0 assembles."""
    data = corpus(n, 5)
    nblocks = (n + block_size - 1) // block_size
    per = (nblocks + ranks - 1) // ranks
    c = K.Codec("NONE", entropy, block_size, lib=be.lib)
    segs, bits, keep = [], [], []
    for r in range(ranks):
        lo, hi = min(r * per, nblocks) * block_size, min(min((r + 1) * per, nblocks) * block_size, n)
        part = data[lo:hi]
        cap = len(part) + len(part) // 2 + 65536
        dst, kdst = be.empty(cap)
        if part:
            src, ksrc = be.to_dev(part)
            keep.append(ksrc)
            nb = c.dev_compress_blocks(src, len(part), dst, cap)
            back, kback = be.empty(len(part) + 64)
            assert c.dev_decompress_blocks(dst, nb, back, len(part) + 64) == len(part)   # each rank decodes its own segment
            assert be.to_host(kback, len(part)) == part
        else:
            nb = 0
        segs.append(dst)
        bits.append(nb)
        keep.append(kdst)
    cap = n + n // 2 + 65536
    out, kout = be.to_dev(bytes([0xA5]) * cap)          # stale bytes in the destination: the assembly must not depend on a cleared buffer
    total = c.dev_assemble(n, segs, bits, out, cap)
    assert be.to_host(kout, total) == O.compress(data, "NONE", entropy, block_size)
    c.close()


def check_short_inner_block(be):
    """A stream whose NON-final blocks are short (concatenated segments, other writers): io.Reader accepts it
    (CompressedStream.go:1707-1710 only bounds a block by the stream's block size), so must knz_dev_decompress. Built by
    assembling two independently encoded segments, the first of which ends in a short block."""
    bs = 1 << 14
    for transform, entropy in (("NONE", "HUFFMAN"), ("BWT+RANK+ZRLT", "ANS0")):
        a, b = corpus(bs + 4321, 7), corpus(2 * bs + 99, 8)
        c = K.Codec(transform, entropy, bs, lib=be.lib)
        segs, bits, keep = [], [], []
        for part in (a, b):
            src, ks = be.to_dev(part)
            cap = 2 * len(part) + 65536
            dst, kd = be.empty(cap)
            bits.append(c.dev_compress_blocks(src, len(part), dst, cap))
            segs.append(dst)
            keep += [ks, kd]
        n = len(a) + len(b)
        cap = 2 * n + 65536
        out, kout = be.empty(cap)
        total = c.dev_assemble(n, segs, bits, out, cap)
        stream = be.to_host(kout, total)
        assert O.decompress(stream, n + 64) == a + b                         # the oracle's Reader takes the short block in the middle
        sp, ksp = be.to_dev(stream, 4)
        # blocks are placed at multiples of the block size first, so the destination holds nblocks * block_size
        back, kb = be.empty(5 * bs + 64)
        assert c.dev_decompress(sp, len(stream), back, 5 * bs + 64) == n
        assert be.to_host(kb, n) == a + b
        c.close()


def transform_inputs(zrlt=False, max_len=1 << 30):
    for name, data in _transform_inputs(zrlt):
        if len(data) <= max_len:
            yield name, data


def _transform_inputs(zrlt=False):
    rng = np.random.default_rng(1234)
    yield "A", b"A"
    yield "AA", b"AA"
    yield "AB", b"AB"
    yield "all256", bytes(range(256))
    yield "seq", bytes([0, 1, 2, 2, 2, 2, 7, 9, 9, 16, 16, 16, 1] + [3] * 19)
    a = bytearray([8]) * 80000
    a[0] = 1
    yield "eights80k", bytes(a)
    yield "short", bytes([0, 0, 1, 1, 2, 2, 2, 2, 2, 2, 2, 3, 3, 3])
    r = 5 if zrlt else 100
    for i in range(3, 6):
        v = rng.integers(0, r, 1 << (i + 6))
        v[v >= 33] = 0
        yield f"zeros{i}", v.astype(np.uint8).tobytes()
    for k in range(6):
        out = bytearray(20)
        while len(out) < 1024:
            ln = int(rng.integers(0, 120))
            if ln % 3 == 0 or ln == 0:
                ln = 1
            out += bytes([int(rng.integers(0, 5 if zrlt else 256))]) * ln
        yield f"runs{k}", bytes(out[:1024])
    yield "mississippi", b"mississippi"
    yield "ramp70000", bytes(i & 255 for i in range(70000))
    yield "text", corpus(60000)
    yield "zeros20000", bytes(20000)
    yield "fefe", bytes([0xFE, 0xFF, 0, 0, 0xFF, 1, 0]) * 300 + bytes(900)
    yield "rand30000", rng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
    # LZP: 0xFC flag bytes with and without a prediction, predictions that hold for 64.. / 254+64.. bytes, a repeat at the very end
    fc = bytearray(rng.integers(0, 256, 3000, dtype=np.uint8).tobytes())
    fc[100:110] = bytes([0xFC]) * 10
    rep = bytes(fc[200:200 + 700])
    yield "fcrep", bytes(fc) + rep + bytes([0xFC, 1, 2, 0xFC]) + rep[:90] + bytes([0xFC]) * 70 + rep[:300] + bytes(fc[:64])
    yield "binary", (rng.integers(0, 256, 50000, dtype=np.uint8) & rng.integers(0, 256, 50000, dtype=np.uint8) & rng.integers(0, 256, 50000, dtype=np.uint8)).tobytes()


_TID = {"BWT": 1, "ZRLT": 6, "MTFT": 7, "RANK": 8, "LZ": 3, "LZX": 16, "SRT": 13, "LZP": 14, "UTF": 17, "TEXT": 10}


def utf_text(n, seed, kinds=(0, 1, 2, 3)):
    """UTF-8 text: ASCII, Cyrillic (2 bytes), CJK (3 bytes), emoji (4 bytes) words."""
    r = np.random.default_rng(seed)
    pools = {0: list(range(0x20, 0x7F)), 1: list(range(0x400, 0x450)), 2: list(range(0x4E00, 0x4E00 + 3000)), 3: [0x1F600 + i for i in range(60)]}
    out, total = [], 0
    while total < n:
        k = kinds[int(r.integers(0, len(kinds)))]
        w = "".join(chr(pools[k][int(i)]) for i in r.integers(0, len(pools[k]), int(r.integers(1, 12))))
        b = (w + " ").encode("utf-8")
        out.append(b)
        total += len(b)
    return b"".join(out)[:n]


def utf_inputs():
    yield "mix", utf_text(50000, 1)
    yield "cyr", utf_text(30000, 2, (0, 1, 1, 1))
    yield "ascii", utf_text(20000, 3, (0,))                             # declined: under 1/8 continuation bytes
    yield "cjk", utf_text(120000, 4, (2, 2, 0))
    yield "bomlike", b"x\xef\xbb\xbf" + utf_text(5000, 5, (1,))          # the BOM test looks at bytes 1..3
    yield "trunc1", utf_text(40000, 6, (1, 2))[1:]                      # starts inside a code point
    yield "trunc2", utf_text(40000, 7, (2, 3))[2:-1]                    # ... and ends inside one
    yield "small", utf_text(1024, 8, (1,))
    yield "tiny", utf_text(1000, 9, (1,))                               # under the minimum block size
    bad = bytearray(utf_text(30000, 10, (1, 2)))
    bad[15000] = 0xC0
    yield "badbyte", bytes(bad)
    bad = bytearray(utf_text(30000, 11, (1, 2)))
    bad[15001] = 0x41 if bad[15000] >= 0xC2 else bad[15001]            # (possibly) a lead byte without its continuation
    yield "badpair", bytes(bad)
    bad = bytearray(utf_text(30000, 12, (2,)))
    for i in range(len(bad) - 3, 100, -1):
        if bad[i] >= 0xE0:
            bad[i + 2] = 0x20                                           # third byte of a 3-byte sequence
            break
    yield "badthird", bytes(bad)
    # a two-byte lead byte at the last position of the region (count - 5) with anything behind it: the reference checks pairs inside the region
    # only and packs the two bytes as they are (UTFCodec.go:459-499, :521-546); three shapes of the byte behind it
    for k, nxt in enumerate((0x41, 0x95, 0xD1)):
        body = utf_text(20000, 15 + k, (1,)).decode("utf-8", errors="ignore").encode("utf-8")     # (ends on a code point boundary)
        yield "lastlead%d" % k, body + b"  " + bytes([0xD0, nxt]) + b"abc"
    yield "magic", bytes([0x1F, 0x8B]) + utf_text(30000, 13, (1,))      # gzip magic (only through the stream API)
    yield "binary", np.random.default_rng(14).integers(0, 256, 20000, dtype=np.uint8).tobytes()


def check_transform(be, tname, max_len=1 << 30):
    """kanzi.ByteTransform objects through knz_transform_forward / knz_transform_inverse vs the oracle."""
    c = K.Codec("NONE", "NONE", 1 << 20, lib=be.lib)
    t = K.ByteTransform(c, tname)
    tid = _TID[tname]
    applied = 0
    for name, data in (utf_inputs() if tname == "UTF" else transform_inputs(zrlt=(tname == "ZRLT"), max_len=max_len)):
        g = t.forward(data)
        o = O.transform_forward(tid, data)
        assert (g is None) == (o is None), (tname, name)
        if o is None:
            continue
        assert g == o, (tname, name, len(g), len(o))
        applied += 1
        back = t.inverse(o, len(data) + max(512, len(data) >> 4))
        assert back == data, (tname, name)
    assert applied >= (5 if tname == "UTF" else 6)
    c.close()


def lz_inverse_inputs(big=False):
    """Inputs whose LZ form carries every kind of length record: literal runs of 7+, 261+ and 65797+ bytes (1, 3 and 4 extension bytes,
    LZCodec.go:193-212), match lengths that need 1- and 3-byte extensions, long chains of repeat distances, overlapping matches."""
    rng = np.random.default_rng(99)
    rnd = rng.integers(0, 256, 70100, dtype=np.uint8).tobytes()
    yield "lit4", rnd + rnd[1000:31000] + bytes(70000) + corpus(30000)
    yield "lit3", rnd[:300] + rnd[:300] + rnd[300:2000] + rnd[100:1500] + bytes(600) + rnd[2000:2300] * 7
    per = rnd[:37]
    yield "periodic", per * 4000 + rnd[:64] + per * 100 + bytes([1, 2, 3]) * 5000
    rec = bytearray(corpus(400))
    out = bytearray()
    for i in range(600 if big else 150):                                       # records with small edits: repeat distances all the time
        r = bytearray(rec)
        r[int(rng.integers(0, 400))] = int(rng.integers(0, 256))
        out += r
    yield "records", bytes(out)
    yield "text", corpus(600000 if big else 90000, seed=5)
    # long literal runs in the MIDDLE of many short ones: the walk over the literal-length extensions takes sixteen of them as one-byte forms and
    # walks the sixteen again when one was longer (lz_inv_par.hip), and its cursor windows jump
    txt = corpus(40000, seed=6)
    parts = []
    for i in range(24):
        parts.append(txt[i * 1500: (i + 1) * 1500])
        parts.append(rng.integers(0, 256, int(rng.integers(255, 420)) if i != 11 else 66100, dtype=np.uint8).tobytes())
    yield "lit_mixed", b"".join(parts)


def check_lz_inverse_forms(be, monkeypatch, big=False):
    """LZ / LZX inverse: the parallel form (lz_inv_par.hip, default) and the one-wave kernel (lz.hip, KNZ_LZ_INV_CHAIN) against the oracle's
    forward output; well-formed blocks never reach the one-wave kernel (KNZ_COUNTER_LZ_INV_SERIAL_BLOCKS), damaged ones do and both
    forms agree with the oracle on error-or-bytes."""
    c = K.Codec("NONE", "NONE", 1 << 20, lib=be.lib)
    rng = np.random.default_rng(5)
    for tname in ("LZ", "LZX"):
        t = K.ByteTransform(c, tname)
        tid = _TID[tname]
        for name, data in lz_inverse_inputs(big):
            o = O.transform_forward(tid, data)
            assert o is not None, (tname, name)
            cap = len(data) + max(512, len(data) >> 4)
            monkeypatch.delenv("KNZ_LZ_INV_CHAIN", raising=False)
            assert t.inverse(o, cap) == data, (tname, name, "parallel")
            assert c.last_counter(3) == 0, (tname, name)
            monkeypatch.setenv("KNZ_LZ_INV_CHAIN", "1")
            assert t.inverse(o, cap) == data, (tname, name, "one wave")
            assert c.last_counter(3) == 1, (tname, name)
            monkeypatch.delenv("KNZ_LZ_INV_CHAIN")
            # damaged: flip bytes in the header / token / distance / length regions
            for trial in range(6 if not big else 2):
                bad = bytearray(o)
                for _ in range(int(rng.integers(1, 4))):
                    pos = int(rng.integers(0, 13)) if trial == 0 else int(rng.integers(0, len(bad)))
                    bad[pos] ^= 1 << int(rng.integers(0, 8))
                try:
                    exp = O.transform_inverse(tid, bytes(bad), cap)
                except O.OracleError:
                    exp = None
                got = []
                for env in (None, "1"):
                    if env:
                        monkeypatch.setenv("KNZ_LZ_INV_CHAIN", env)
                    try:
                        got.append(t.inverse(bytes(bad), cap))
                    except K.KnzError:
                        got.append(None)
                    monkeypatch.delenv("KNZ_LZ_INV_CHAIN", raising=False)
                assert got[0] == got[1], (tname, name, trial, "the two device forms disagree on a damaged block")
                if exp is None or got[0] is None:
                    assert exp is None or got[0] is None or got[0] == exp
                else:
                    assert got[0] == exp, (tname, name, trial)
    c.close()


def lz_forward_inputs(big=False):
    """Blocks that make the segment-parallel parse work: long stretches without matches (the miss acceleration jumps over positions:
    holes), matches found from inside such stretches, repeat distances across segment borders, text, records."""
    import bench_corpus as bc
    rng = np.random.default_rng(17)
    n = 600_000 if big else 24_000
    yield "exe", bc._segment("exe", n, 2).tobytes()
    yield "records", bc._segment("records", n // 2, 4).tobytes()
    yield "mix", (bc._segment("exe", n // 3, 5).tobytes() + rng.integers(0, 256, n // 4, dtype=np.uint8).tobytes() + bc._segment("text", n // 4, 7).tobytes() +
                  bytes(n // 16) + rng.integers(0, 256, n // 8, dtype=np.uint8).tobytes() + bc._segment("text", n // 8, 7).tobytes())
    # incompressible stretches longer than two (small) segments with matches right behind them: a parse that skips runs over whole segments,
    # which turns segments that recorded a trace from a guessed state into pass-through ones (their traces must leave the hole maps)
    txt = bc._segment("text", 3000, 9).tobytes()
    yield "overshoot", b"".join(rng.integers(0, 256, int(k), dtype=np.uint8).tobytes() + txt[: int(m)] + txt[: int(m)]
                                for k, m in zip(rng.integers(700, 3000, 12), rng.integers(40, 900, 12)))
    if big:
        yield "img16", bc._segment("img16", n // 2, 3).tobytes()
        yield from lz_inverse_inputs(big)
    else:
        per = rng.integers(0, 256, 37, dtype=np.uint8).tobytes()
        yield "periodic", per * 300 + rng.integers(0, 256, 64, dtype=np.uint8).tobytes() + per * 40 + bytes([1, 2, 3]) * 900


def check_lz_forward_forms(be, monkeypatch, big=False, segs=(256, 512, 1024)):
    """LZ / LZX forward: the segment-parallel parse (lz_fwd_seg.hip, default) with small segments so that every input spans many of them,
    the one-wave table-free parse (lz_par.hip, KNZ_LZ_ONE_WAVE) and the first form (lz.hip, KNZ_LZ_CHAIN): all three == the oracle, and the
    segment-parallel one settles on its own (KNZ_COUNTER_LZ_FWD_SERIAL_BLOCKS == 0)."""
    c = K.Codec("NONE", "NONE", 4 << 20, lib=be.lib)
    keys = ("KNZ_LZ_SEG", "KNZ_LZ_ONE_WAVE", "KNZ_LZ_CHAIN", "KNZ_LZS_WAVES", "KNZ_LZS_CHG_CAP")
    for tname in ("LZ", "LZX"):
        t = K.ByteTransform(c, tname)
        tid = _TID[tname]
        for name, data in lz_forward_inputs(big):
            o = O.transform_forward(tid, data)
            # the segment-parallel parse in both of its forms: one lane per segment (default since round 6) and one wave per segment (KNZ_LZS_WAVES)
            envs = [("KNZ_LZ_SEG", str(sg)) for sg in segs] + [("KNZ_LZ_SEG", str(sg), "KNZ_LZS_WAVES", "1") for sg in segs]
            # (the list of moved map words holds a block's whole map; a list of 16 words makes "too many moved: everybody runs again" the usual case)
            envs += [("KNZ_LZ_SEG", str(segs[0]), "KNZ_LZS_CHG_CAP", "16")]
            if big or tname == "LZ":
                envs += [("KNZ_LZ_SEG", ""), ("KNZ_LZ_SEG", "", "KNZ_LZS_WAVES", "1"), ("KNZ_LZ_ONE_WAVE", "1"), ("KNZ_LZ_CHAIN", "1")]
            for env in envs:
                for k in keys:
                    monkeypatch.delenv(k, raising=False)
                for k, v in zip(env[0::2], env[1::2]):
                    if v:
                        monkeypatch.setenv(k, v)
                g = t.forward(data)
                assert g == o, (tname, name, env)
                if env[0] == "KNZ_LZ_SEG":
                    # (records locked onto different repeat distances settle one segment per round: with tiny segments such a block may run
                    # out of rounds and go to the one-wave kernel, which is exact too; at the default segment size everything here settles)
                    if not env[1]:
                        assert c.last_counter(4) == 0, (tname, name, env, "left to the one-wave kernel")
                    assert 1 <= c.last_counter(5) <= 256, (tname, name, env, c.last_counter(5))
    for k in keys:
        monkeypatch.delenv(k, raising=False)
    c.close()


def check_rank_chain_variants(be, monkeypatch, max_len=1 << 30, bwt_len=150000):
    """Inverse RANK chain (rank_inv.hip): every kept variant of the step x the packed / three-register forms (the latter is
    what blocks > 8 MiB use), on inputs with ranks >= 64, all-zero words, ragged tails; against the oracle's forward."""
    import os
    c = K.Codec("NONE", "NONE", 1 << 20, lib=be.lib)
    t = K.ByteTransform(c, "RANK")
    rng = np.random.default_rng(77)
    extra = [("tail%d" % k, (np.minimum(rng.geometric(0.3, 4099 + k), 200) - 1).astype(np.uint8).tobytes()) for k in range(4)]
    extra.append(("bwtlike", O.transform_forward(_TID["BWT"], corpus(bwt_len))))
    cases = [(nm, d) for nm, d in list(transform_inputs(max_len=max_len)) + extra if len(d) > 0]
    fwd = [(nm, d, O.transform_forward(_TID["RANK"], d)) for nm, d in cases]
    for variant in ("0", "5"):                       # 0 = the round-1 kernel (cross-check), 5 = the default; -DKNZ_MEASURE builds carry the earlier forms as 3, 4, 6..8
        for unpacked in (False, True, "cut"):
            monkeypatch.setenv("KNZ_RANK_VARIANT", variant)
            monkeypatch.delenv("KNZ_RANK_UNPACKED", raising=False)
            monkeypatch.delenv("KNZ_RANK_CUT", raising=False)
            if unpacked is True:
                monkeypatch.setenv("KNZ_RANK_UNPACKED", "1")
            elif unpacked == "cut":                                  # packed up to rank 1024, the three-register form behind it (what an
                if variant not in ("4", "5", "6"):                        # 8 MiB block behind a BWT does at rank 2^23)
                    continue
                monkeypatch.setenv("KNZ_RANK_CUT", "1024")
            for nm, d, f in fwd:
                if f is None:
                    continue
                assert t.inverse(f, len(d) + 512) == d, (variant, unpacked, nm)
    monkeypatch.delenv("KNZ_RANK_VARIANT", raising=False)
    monkeypatch.delenv("KNZ_RANK_UNPACKED", raising=False)
    monkeypatch.delenv("KNZ_RANK_CUT", raising=False)
    c.close()


def rank_patterns(scale=1):
    """Rank sequences (any byte sequence is a valid input of SBRT.Inverse, SBRT.go:180-226) aimed at the paths of the device's hand-written
    blocks (rank_inv_asm.h): ranks on both sides of every register boundary of the list (63/64, 127/128, 191/192), the last rank, words of
    four zeros between high ranks, whole groups of zeros, groups with exactly one high rank at each of the 16 positions, and lengths that end
    inside a word, a group and a row of 64."""
    rng = np.random.default_rng(4242)
    n0 = 4096 * scale
    yield "uniform", rng.integers(0, 256, n0 + 37).astype(np.uint8).tobytes()
    for lo, hi in ((63, 64), (127, 128), (191, 192), (254, 255), (0, 255), (0, 64), (1, 192)):
        yield "pair%d_%d" % (lo, hi), rng.choice(np.array([lo, hi], dtype=np.uint8), n0 + 5).tobytes()
    yield "all255", bytes([255]) * (n0 + 1)
    yield "all64", bytes([64]) * (n0 + 2)
    v = np.zeros(n0 + 3, dtype=np.uint8)
    v[rng.integers(0, len(v), len(v) // 23)] = rng.integers(64, 256, len(v) // 23).astype(np.uint8)
    yield "zeros_with_high", v.tobytes()
    v = np.minimum(rng.geometric(0.25, n0 + 61), 63).astype(np.uint8) - 1
    for k in range(16):
        v[16 * (3 * k + 1) + k] = 64 + 12 * k          # one high rank per group, at every position of a group
    yield "one_high_per_group", v.tobytes()
    v = rng.integers(0, 256, n0 + 19).astype(np.uint8)
    v[64:64 + 640] = 0                                   # whole groups / rows of zeros inside random ranks
    v[1000:1004] = 0
    yield "uniform_with_zero_rows", v.tobytes()
    for n in (1, 3, 4, 5, 15, 16, 17, 63, 64, 65, 127, 129):
        yield "len%d" % n, rng.integers(0, 256, n).astype(np.uint8).tobytes()


def check_rank_inverse_patterns(be, scale=1):
    """Device SBRT RANK inverse == oracle inverse on rank_patterns(), and forward of the result gives the ranks back."""
    c = K.Codec("NONE", "NONE", 1 << 20, lib=be.lib)
    t = K.ByteTransform(c, "RANK")
    for nm, ranks in rank_patterns(scale):
        want = O.transform_inverse(_TID["RANK"], ranks, len(ranks) + 64)
        got = t.inverse(ranks, len(ranks) + 512)
        assert got == want, nm
        assert t.forward(got) == ranks, nm
    c.close()


def check_bwt_list_ranking(be, monkeypatch, max_len=70000):
    """Inverse BWT by list ranking (bwt.hip: splitter walks, Wyllie ranking, emit) forced onto small blocks (by default blocks
    below 64 KiB take the 8-chain kernel): streams with ragged last blocks, the BWT_test.go inputs, and damaged streams, which the
    ranking must hand back to the chains kernel (error-or-bytes agreement with the oracle is checked by check_corrupt_streams)."""
    monkeypatch.setenv("KNZ_BWT_RANK_MIN", "256")
    for cfg in (("BWT", "NONE", 1 << 14, 5 * (1 << 14) + 777), ("BWT+RANK+ZRLT", "ANS1", 1 << 15, 100003), ("BWT", "HUFFMAN", 1024, 5000),
                ("BWT+MTFT+ZRLT", "ANS0", 4096, 4096 * 3 + 257)):
        check_stream(be, *cfg)
    c = K.Codec("NONE", "NONE", 1 << 20, lib=be.lib)
    t = K.ByteTransform(c, "BWT")
    for name, data in transform_inputs(max_len=max_len):
        if len(data) < 2:
            continue
        f = O.transform_forward(_TID["BWT"], data)
        assert t.inverse(f, len(data) + 512) == data, name
    c.close()
    monkeypatch.delenv("KNZ_BWT_RANK_MIN", raising=False)


def reference_test_inputs():
    """tests/golden/reference_inputs.json: the deterministic input literals of the reference's own tests, extracted from the
    reference source by tests/golden/make_reference_inputs.py (not re-typed)."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_inputs.json")))
    return ([(e["name"], bytes.fromhex(e["hex"])) for e in g["entropy"]], [(e["name"], bytes.fromhex(e["hex"])) for e in g["transform"]])


def check_reference_inputs(be):
    """Every input literal of the reference's own tests (Entropy_test.go:617-693, Transforms_test.go:165-258,534-601,
    BWT_test.go:60-84) through both directions of the boundary, crossed with the oracle: device-encode -> oracle-decode and
    oracle-encode -> device-decode, as codec objects and as whole streams. (The reference's tests hold no output bytes: this pins
    the inputs it exercises, not the bytes a Go build would write.)"""
    ent, trf = reference_test_inputs()
    c = K.Codec("NONE", "NONE", 1 << 16, lib=be.lib)
    for ename in ("HUFFMAN", "ANS0", "ANS1", "FPAQ", "NONE"):
        et = O.entropy_type(ename)
        enc, dec = K.EntropyEncoder(c, ename), K.EntropyDecoder(c, ename)
        for name, data in ent + trf:
            if len(data) == 0 or len(data) > 5000:
                continue
            if ename == "ANS1" and len(data) in (2, 3):
                continue                                   # the reference panics on an order-1 chunk of 2-3 bytes (ANSRangeCodec.go:353-362): mirrored, tested elsewhere
            gb, gbits = enc.write(data)
            ob, obits = O.entropy_encode(et, data)
            assert (gb, gbits) == (ob, obits), (ename, name)
            assert O.entropy_decode(et, gb, len(data))[0] == data, (ename, name, "device-encode -> oracle-decode")
            assert dec.read(ob, len(data))[0] == data, (ename, name, "oracle-encode -> device-decode")
    for tname in ("BWT", "RANK", "MTFT", "ZRLT", "LZ", "LZX", "LZP", "SRT", "TEXT", "UTF"):
        t = K.ByteTransform(c, tname)
        tid = _TID[tname]
        for name, data in trf + ent:
            if len(data) == 0 or len(data) > (1 << 16):
                continue
            g = t.forward(data)
            O.set_ctx(1 << 16, O.E_NONE)                                     # (TEXT reads ctx: the handle's block size and entropy stage)
            o = O.transform_forward(tid, data)
            assert (g is None) == (o is None), (tname, name)
            if o is None:
                continue
            assert g == o, (tname, name)
            assert O.transform_inverse(tid, g, len(data) + 1024) == data, (tname, name, "device-forward -> oracle-inverse")
            assert t.inverse(o, len(data) + max(512, len(data) >> 4)) == data, (tname, name, "oracle-forward -> device-inverse")
    c.close()
    # whole streams, both ways (block size 64 KiB: the 80 000-byte input spans two blocks)
    for transform, entropy in (("NONE", "HUFFMAN"), ("BWT+RANK+ZRLT", "ANS1"), ("LZ", "ANS0"), ("BWT+RANK+ZRLT", "FPAQ"), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0")):
        cs = K.Codec(transform, entropy, 1 << 16, lib=be.lib)
        for name, data in trf + ent:
            if len(data) == 0:
                continue
            src, ks = be.to_dev(data)
            cap = 2 * len(data) + (1 << 18)
            dst, kd = be.empty(cap)
            nb = cs.dev_compress(src, len(data), dst, cap)
            got = be.to_host(kd, nb)
            exp = O.compress(data, transform, entropy, 1 << 16)
            assert got == exp, (transform, entropy, name)
            assert O.decompress(got, len(data) + 64) == data, (transform, entropy, name, "device stream -> oracle reader")
            s2, k2 = be.to_dev(exp)
            out, ko = be.empty(len(data) + 64)
            assert cs.dev_decompress(s2, len(exp), out, len(data) + 64) == len(data)
            assert be.to_host(ko, len(data)) == data, (transform, entropy, name, "oracle stream -> device reader")
        cs.close()


def check_rank_pipe(be, monkeypatch, sizes=((50000, 1 << 14), (300000, 1 << 16), (1000, 1024), (200001, 1 << 17)), seeds=(5, 6), seqs=("BWT+RANK+ZRLT", "RANK+ZRLT"),
                    forms=("pipe", "regular", "unpacked", "cut", "two_groups")):
    """Decode of ...RANK+ZRLT / ANS1: the two inverses run as one chain under the rANS decoder (rank_pipe.hip). Same bytes as the regular stage kernels
    (KNZ_NO_RANK_PIPE), every coded block taken by the chain (knz_last_counter 6), the three-register form and a moved packed / unpacked cut included;
    inputs with long zero runs (digits across lane and piece boundaries), escapes (0xFE / 0xFF ranks) and blocks a stage skips."""
    def shapes(n, seed):
        r = np.random.default_rng(seed)
        yield "corpus", corpus(n, seed)
        yield "zeros", bytes(n)
        yield "sparse", np.where(r.random(n) < 0.002, r.integers(1, 256, n), 0).astype(np.uint8).tobytes()
        yield "random", r.integers(0, 256, n, dtype=np.uint8).tobytes()                     # (ranks up to 255: escapes in the ZRLT stream; ZRLT may be skipped)
        yield "runs", b"".join(bytes([int(r.integers(0, 256))]) * int(r.integers(1, 5000)) for _ in range(max(n // 2000, 2)))[:n] or b"x"
        a = np.zeros(n, dtype=np.uint8)
        a[:: max(n // 300, 1)] = r.integers(1, 256, len(a[:: max(n // 300, 1)]))
        yield "steps", a.tobytes()
    for seq in seqs:
        for n, bs in sizes:
            for seed in seeds:
                for name, data in shapes(n, seed):
                    n2 = len(data)
                    stream = O.compress(data, seq, "ANS1", bs)
                    sp, k1 = be.to_dev(stream)
                    res = {}
                    for form in forms:
                        for v in ("KNZ_NO_RANK_PIPE", "KNZ_RANK_UNPACKED", "KNZ_RANK_CUT", "KNZ_RANK_PIPE_TWO_GROUPS"):
                            monkeypatch.delenv(v, raising=False)
                        if form == "regular":
                            monkeypatch.setenv("KNZ_NO_RANK_PIPE", "1")
                        elif form == "unpacked":
                            monkeypatch.setenv("KNZ_RANK_UNPACKED", "1")
                        elif form == "cut":
                            monkeypatch.setenv("KNZ_RANK_CUT", "512")
                        elif form == "two_groups":                                         # (the long chains in a launch of their own, the stages behind the chain in two passes)
                            monkeypatch.setenv("KNZ_RANK_PIPE_TWO_GROUPS", "1")
                        c = K.Codec(seq, "ANS1", bs, lib=be.lib)
                        out, ko = be.empty(n2 + 4096)
                        nd = c.dev_decompress(sp, len(stream), out, n2 + 4096)
                        assert nd == n2 and be.to_host(ko, nd) == data, (seq, n, bs, seed, name, form)
                        res[form] = c.last_counter(6)
                        c.close()
                    assert res["regular"] == 0
                    assert len({v for k, v in res.items() if k != "regular"}) == 1, res
                    if name in ("corpus", "sparse", "steps"):
                        assert res["pipe"] >= 1, (seq, n, bs, seed, name, "no block took the fused chain")
    for v in ("KNZ_NO_RANK_PIPE", "KNZ_RANK_UNPACKED", "KNZ_RANK_CUT", "KNZ_RANK_PIPE_TWO_GROUPS"):
        monkeypatch.delenv(v, raising=False)


def check_device_vs_ref(be, quick=False):
    """The device against oracle/_ref DIRECTLY (the reference's own sources, translated and compiled: tests/ref_lib.py), no hand-written oracle in
    between: entropy codec objects and transform objects in both directions, then whole streams written by the reference's Writer and read by
    the device, and the device's streams read by the reference's Reader."""
    import ref_lib as R
    ent, trf = reference_test_inputs()
    extra = [] if quick else [(n, d) for n, d in entropy_inputs()]           # (quick: the emulator build runs the reference's own test inputs only)
    c = K.Codec("NONE", "NONE", 1 << 16, lib=be.lib)
    for ename in ("HUFFMAN", "ANS0", "ANS1", "FPAQ", "NONE"):
        et = R.entropy_type(ename)
        enc, dec = K.EntropyEncoder(c, ename), K.EntropyDecoder(c, ename)
        for name, data in ent + trf + extra:
            if len(data) == 0 or (ename == "ANS1" and len(data) in (2, 3)):
                continue
            gb, gbits = enc.write(data)
            rb, rbits = R.entropy_encode(et, data)
            assert (gb, gbits) == (rb, rbits), (ename, name, "device encode != reference encode")
            assert R.entropy_decode(et, gb, len(data))[0] == data, (ename, name, "device-encode -> reference-decode")
            assert dec.read(rb, len(data))[0] == data, (ename, name, "reference-encode -> device-decode")
    tin = [] if quick else [(n, d) for n, d in transform_inputs(max_len=1 << 16)] + list(utf_inputs()) + [(n, d) for n, d in text_inputs(50000)]
    for tname in ("BWT", "RANK", "MTFT", "ZRLT", "LZ", "LZX", "LZP", "SRT", "TEXT", "UTF"):
        t = K.ByteTransform(c, tname)
        tid = _TID[tname]
        for name, data in trf + ent + tin:
            if len(data) == 0 or len(data) > (1 << 16):
                continue
            g = t.forward(data)
            R.set_ctx(1 << 16, R.entropy_type("NONE"))                      # (TEXT reads ctx: the handle's block size and entropy stage)
            r = R.transform_forward(tid, data)
            assert (g is None) == (r is None), (tname, name, "one declines, the other does not")
            if r is None:
                continue
            assert g == r, (tname, name, "device forward != reference forward")
            assert R.transform_inverse(tid, g, len(data) + 1024) == data, (tname, name, "device-forward -> reference-inverse")
            assert t.inverse(r, len(data) + max(512, len(data) >> 4)) == data, (tname, name, "reference-forward -> device-inverse")
    c.close()
    for transform, entropy, ck in (("NONE", "HUFFMAN", 0), ("BWT+RANK+ZRLT", "ANS1", 64), ("LZ", "ANS0", 32), ("BWT+RANK+ZRLT", "FPAQ", 0),
                                   ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0", 0)):
        cs = K.Codec(transform, entropy, 1 << 16, ck, lib=be.lib)
        for name, data in (trf[:6] + ent[:6] if quick else trf + ent + [("corpus", corpus(200001))]):
            if len(data) == 0:
                continue
            src, ks = be.to_dev(data)
            cap = 2 * len(data) + (1 << 18)
            dst, kd = be.empty(cap)
            nb = cs.dev_compress(src, len(data), dst, cap)
            got = be.to_host(kd, nb)
            exp = R.compress(data, transform, entropy, 1 << 16, ck)
            assert got == exp, (transform, entropy, name, "device stream != the reference Writer's stream")
            assert R.decompress(got, len(data) + 64) == data, (transform, entropy, name, "device stream -> reference Reader")
            s2, k2 = be.to_dev(exp)
            out, ko = be.empty(len(data) + 64)
            assert cs.dev_decompress(s2, len(exp), out, len(data) + 64) == len(data)
            assert be.to_host(ko, len(data)) == data, (transform, entropy, name, "reference stream -> device reader")
        cs.close()


def _huffman_header_with_wrapped_delta(payload, nbits):
    """Re-encodes one negative code-length delta (-2..-11) of a Huffman chunk header with the 16-bit Exp-Golomb form whose
    magnitude wraps as int8 (ExpGolombCodec.go:159-190, readLengths casts to int8): a stream no kanzi encoder writes but
    every kanzi decoder accepts. Returns (payload', nbits') or None when the header holds no such delta."""
    bits = "".join(f"{b:08b}" for b in payload)[:nbits]
    pos = 0
    if bits[0] == "0":
        assert bits[1] == "0"
        count, pos = 256, 2
    else:
        last = int(bits[1:6], 2)
        pos = 6
        count = 0
        for _ in range(last + 1):
            count += bits[pos:pos + 8].count("1")
            pos += 8
    for _ in range(count):
        if bits[pos] == "1":
            pos += 1
            continue
        z = 0
        while bits[pos + z] == "0":
            z += 1
        val = int(bits[pos + z + 1: pos + 2 * z + 2], 2)
        mag = (val >> 1) + (1 << z) - 1
        if (val & 1) and 2 <= mag <= 11:
            wrapped = "0000000" + "1" + f"{((256 - mag - 127) << 1):08b}"    # res = 256 - mag, int8(res) = -mag
            nb = bits[:pos] + wrapped + bits[pos + 2 * z + 2:]
            out = bytes(int(nb[i:i + 8].ljust(8, "0"), 2) for i in range(0, len(nb), 8))
            return out, len(nb)
        pos += 2 * z + 2
    return None


def check_huffman_shapes(be):
    """Inputs that drive the parallel Huffman decoder down its less common paths: many synchronisation rounds
    (incompressible bytes), more symbols in a lane than its LDS row holds (long runs of a 1-bit code between 9-bit codes),
    and a header only the serial parser takes (int8-wrapped Exp-Golomb delta)."""
    rng = np.random.default_rng(11)
    n = 5 * 16384 + 1234
    runs = np.empty(n, dtype=np.uint8)
    for i in range(0, n, 1024):
        runs[i:i + 512] = 65
        runs[i + 512:i + 1024] = rng.integers(0, 256, len(runs[i + 512:i + 1024]), dtype=np.uint8)
    cases = [("rand", rng.integers(0, 256, n, dtype=np.uint8).tobytes()), ("runs", runs.tobytes())]
    for name, data in cases:
        c = K.Codec("NONE", "HUFFMAN", 1 << 16, lib=be.lib)
        exp = O.compress(data, "NONE", "HUFFMAN", 1 << 16)
        sp, ks = be.to_dev(exp, 4)
        out, kout = be.empty(len(data) + 64)
        nd = c.dev_decompress(sp, len(exp), out, len(data) + 64)
        assert nd == len(data) and be.to_host(kout, nd) == data, name
        assert c.last_counter(0) == 0, name
        c.close()
    # the header the parallel parser must hand back
    c = K.Codec("NONE", "HUFFMAN", 1 << 16, lib=be.lib)
    dec = K.EntropyDecoder(c, "HUFFMAN")
    et = O.entropy_type("HUFFMAN")
    done = False
    for seed in range(8):
        data = np.minimum(np.random.default_rng(seed).geometric(0.03, 9000), 255).astype(np.uint8).tobytes()
        ob, obits = O.entropy_encode(et, data)
        alt = _huffman_header_with_wrapped_delta(ob, obits)
        if alt is None:
            continue
        ab, abits = alt
        assert O.entropy_decode(et, ab, len(data))[0] == data          # the reference-order decoder accepts it
        dd, used = dec.read(ab, len(data))
        assert dd == data and used == abits
        assert c.last_counter(0) == 1                                   # ... and the device took its serial path for it
        done = True
        break
    assert done, "no negative delta found in 8 headers"
    c.close()


def check_checksums(be):
    """-x / --checksum: XXHash32/64 of every block on the device (stream bit-exact vs the oracle, verified on decode,
    a flipped payload bit is reported as ERR_CRC_CHECK = 19 like decodingTask.decode)."""
    n, bs = 3 * 65536 + 4321, 65536
    data = corpus(n, 21)
    for bits in (32, 64):
        for transform, entropy in (("NONE", "HUFFMAN"), ("RANK+ZRLT", "ANS0")):
            c = K.Codec(transform, entropy, bs, checksum_bits=bits, lib=be.lib)
            src, ks = be.to_dev(data)
            cap = 2 * n + 262144
            dst, kd = be.empty(cap)
            nb = c.dev_compress(src, n, dst, cap)
            got = be.to_host(kd, nb)
            exp = O.compress(data, transform, entropy, bs, checksum_bits=bits)
            assert got == exp, (bits, transform)
            out, ko = be.empty(n + 64)
            assert c.dev_decompress(dst, nb, out, n + 64) == n
            assert be.to_host(ko, n) == data
            # flip one bit of the last block's payload: the entropy decoder still runs, the checksum catches it
            bad = bytearray(exp)
            bad[len(bad) - 40] ^= 0x10
            sp, ksp = be.to_dev(bytes(bad), 4)
            try:
                c.dev_decompress(sp, len(bad), out, n + 64)
                raised = None
            except K.KnzError as e:
                raised = e.code
            assert raised in (19, 13), raised       # CRC mismatch (or a payload the entropy decoder already rejects)
            c.close()
    # batch hook: per-block checksum value
    c = K.Codec("NONE", "HUFFMAN", bs, checksum_bits=32, lib=be.lib)
    bb = K.BlockBatch(c)
    for blocks in ([corpus(bs, 30), corpus(1000, 31)], [corpus(bs, 32), b"tiny"]):
        res = bb.encode(blocks)
        for blk, r in zip(blocks, res):
            o = O.encode_block(blk, O.transform_type("NONE"), O.entropy_type("HUFFMAN"), 32)
            assert r[0] == o["bits"] and r[1] == o["written"]
        assert bb.decode([r[0] for r in res]) == blocks
    c.close()


def check_ans1_table_decoder(be):
    """The order-1 rANS decoder has two forms (LDS-resident cumulated frequencies for small batches, slot tables in HBM for
    large ones); KNZ_ANS1_TABLE_DECODER forces the second."""
    import os
    os.environ["KNZ_ANS1_TABLE_DECODER"] = "1"
    try:
        check_stream(be, "NONE", "ANS1", 1 << 16, 3 * 65536 + 777)
        check_entropy_encode(be, "ANS1")
    finally:
        del os.environ["KNZ_ANS1_TABLE_DECODER"]
    # the LDS decoder's loop exists three times on the device: a hand-written gfx950 block with one LDS round trip per step (default: the pair of the
    # second level, the state computed in every lane, a DPP hand-over inside the state's sixteen lanes), round 4's block with two (KNZ_ANS1_LOHI_LDS) and
    # the compiler's schedule of the same steps (KNZ_ANS1_PLAIN; the only one the emulator runs): all against the oracle on ragged chunk lengths
    # (tiles of 256 steps + tails of 1..3)
    for form in (None, "KNZ_ANS1_PLAIN", "KNZ_ANS1_LOHI_LDS"):
        if form is not None and be.name != "gpu":
            continue                                  # (the emulator build has the compiler's loop only: one pass covers it)
        if form:
            os.environ[form] = "1"
        try:
            for n in (3 * 65536 + 777, 4 * 1024 + 1, 4 * 1024 + 2, 4 * 1024 + 3, 1021):
                check_stream(be, "NONE", "ANS1", 1 << 16, n)
            check_stream(be, "BWT+RANK+ZRLT", "ANS1", 1 << 18, (1 << 18) + 12345)
        finally:
            if form:
                os.environ.pop(form, None)


def check_huffman_split_walk(be):
    """The Huffman decoder has two forms: chunk walk and chunk decoders in one launch (default), or the walk as its own
    launch before the decoders; KNZ_HUF_SPLIT_WALK forces the second."""
    import os
    os.environ["KNZ_HUF_SPLIT_WALK"] = "1"
    try:
        check_stream(be, "NONE", "HUFFMAN", 1 << 16, 3 * 65536 + 777)
        check_stream(be, "NONE", "HUFFMAN", 4096, 4096 * 3 + 15)
        check_block_batch(be, "NONE", "HUFFMAN", 1 << 16, 3, 12345)
    finally:
        del os.environ["KNZ_HUF_SPLIT_WALK"]


def check_golden_streams(be):
    """The device writes the streams of tests/golden/oracle_streams.json (sha256 + length committed with the script that made
    them) - the same inputs the oracle regression test uses, compared without running the oracle."""
    import hashlib, json, os, sys
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, g)
    import make_oracle_vectors as M
    want = json.load(open(os.path.join(g, "oracle_streams.json")))["streams"]
    for t, e, bs, n, ck, sk in M.CASES:
        data = M.data_for(t, n)
        c = K.Codec(t, e, bs, checksum_bits=ck, lib=be.lib, skip_blocks=sk)
        src, ks = be.to_dev(data)
        cap = 2 * n + 262144 * (n // bs + 2)
        dst, kd = be.empty(cap)
        nb = c.dev_compress(src, n, dst, cap)
        got = be.to_host(kd, nb)
        w = want[f"{t}|{e}|{bs}|{n}|{ck}|{int(sk)}"]
        assert (len(got), hashlib.sha256(got).hexdigest()) == (w["len"], w["sha256"]), (t, e, bs, n, ck, sk)
        c.close()


def check_mtft_segments(be):
    """MTFT inverse over several 8 KiB segments (per-segment permutations composed per block) and its one-wave form
    (KNZ_MTFT_CHAIN)."""
    import os
    r = np.random.default_rng(9)
    data = (np.minimum(r.geometric(0.15, 3 * 8192 + 17), 255).astype(np.uint8) ^ (np.arange(3 * 8192 + 17) >> 11).astype(np.uint8)).tobytes()
    c = K.Codec("NONE", "NONE", 1 << 20, lib=be.lib)
    t = K.ByteTransform(c, "MTFT")
    f = t.forward(data)
    assert f == O.transform_forward(_TID["MTFT"], data)
    assert t.inverse(f, len(data) + 512) == data
    os.environ["KNZ_MTFT_CHAIN"] = "1"
    try:
        assert t.inverse(f, len(data) + 512) == data
    finally:
        del os.environ["KNZ_MTFT_CHAIN"]
    c.close()


def check_srt_chain_form(be):
    """SRT forward has two forms: MTFT ranks + stable partition by symbol (default), or the reference's list walk by one wave
    per block; KNZ_SRT_CHAIN forces the second."""
    import os
    os.environ["KNZ_SRT_CHAIN"] = "1"
    try:
        check_transform(be, "SRT", max_len=20000)
        check_stream(be, "BWT+SRT+ZRLT", "ANS0", 1 << 14, 40000)
    finally:
        del os.environ["KNZ_SRT_CHAIN"]


def check_utf_streams(be):
    """UTF stage inside streams: UTF-8 blocks, a block whose magic number sets ctx["dataType"] (the stage declines), UTF twice in
    a sequence (the second stage runs without validateUTF: the sequential-walk form on the device), the -l 5 tail behind it."""
    bs = 1 << 16
    data = (utf_text(bs, 21, (1, 1, 0)) + bytes([0x1F, 0x8B]) + utf_text(bs - 2, 22, (1,)) + b"MZ" + utf_text(bs - 2, 23, (2,)) +
            utf_text(bs, 24, (0,)) + utf_text(bs // 2 + 77, 25, (3, 1)))
    for transform, entropy in (("UTF", "HUFFMAN"), ("UTF", "NONE"), ("UTF+UTF", "ANS0"), ("UTF+BWT+RANK+ZRLT", "ANS0"), ("UTF+LZ", "HUFFMAN")):
        exp = O.compress(data, transform, entropy, bs)
        c = K.Codec(transform, entropy, bs, lib=be.lib)
        src, ks = be.to_dev(data)
        cap = 2 * len(data) + (1 << 20)
        dst, kd = be.empty(cap)
        nb = c.dev_compress(src, len(data), dst, cap)
        assert be.to_host(kd, nb) == exp, (transform, entropy)
        out, ko = be.empty(len(data) + 64)
        assert c.dev_decompress(dst, nb, out, len(data) + 64) == len(data)
        assert be.to_host(ko, len(data)) == data
        c.close()
    assert len(O.compress(data, "UTF", "NONE", bs)) < len(data) - 30000       # the stage applied to the UTF-8 blocks
    # more blocks than one workspace group holds (the UTF stage runs in groups of 64 blocks)
    bs2 = 2048
    many = utf_text(bs2 * 70 + 13, 31, (1, 1, 0))
    for transform, entropy in (("UTF", "NONE"), ("UTF", "HUFFMAN")):
        c = K.Codec(transform, entropy, bs2, lib=be.lib)
        src, ks = be.to_dev(many)
        cap = 2 * len(many) + (1 << 20)
        dst, kd = be.empty(cap)
        nb = c.dev_compress(src, len(many), dst, cap)
        assert be.to_host(kd, nb) == O.compress(many, transform, entropy, bs2), (transform, entropy, "many blocks")
        out, ko = be.empty(len(many) + 64)
        assert c.dev_decompress(dst, nb, out, len(many) + 64) == len(many)
        assert be.to_host(ko, len(many)) == many
        c.close()


def text_inputs(n=200_000):
    import text_corpus as T
    rng = np.random.default_rng(5)
    yield "plain", T.make_text(n, seed=1)
    yield "crlf", T.make_text(n, seed=2, crlf=True)
    yield "utf8", T.make_text(n, seed=3, utf8=0.05)
    yield "markup", T.make_text(n, seed=4, markup=True)
    yield "escapes", T.make_text(n, seed=5, escapes=0.02)
    yield "big-vocabulary", T.make_text(n, seed=6, vocab=40000, static_share=0.05)
    yield "capitals", T.make_text(n, seed=7, upper=0.6)
    yield "long-words", T.make_text(n, seed=8, max_word=40)
    yield "leading-spaces", b"   " + T.make_text(n // 4, seed=9)
    yield "static-words", b"the be and of in to with " * 200
    yield "just-1024", T.make_text(1024, seed=10)
    yield "below-1024", T.make_text(1023, seed=11)
    yield "zip-magic", b"PK\x03\x04" + T.make_text(5000, seed=12)
    yield "dna", rng.choice(np.frombuffer(b"acgt", dtype=np.uint8), 5000).tobytes()
    yield "numeric", rng.choice(np.frombuffer(b"0123456789,. ", dtype=np.uint8), 5000).tobytes()
    yield "base64", rng.choice(np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/", dtype=np.uint8), 5000).tobytes()
    yield "binary", rng.integers(0, 256, 60000).astype(np.uint8).tobytes()
    yield "small-alphabet", rng.choice(np.frombuffer(b"\x01\x02\x03", dtype=np.uint8), 5000).tobytes()
    yield "cyrillic", "".join(chr(0x410 + int(k)) for k in rng.integers(0, 60, 3000)).encode("utf-8")
    yield "bytes-1-199", bytes(range(1, 200)) * 30
    yield "no-gain", (b"qzj xvk wpf " * 100)[:1100]                        # text by the statistics, nothing to replace: the output does not fit
    yield "many-high-bytes", T.make_text(n // 4, seed=13, utf8=0.5)
    if n >= 200_000:                                                       # (GPU runs) more than 16384 words enter the dictionary: 3-letter words stop entering (:801)
        import bench_corpus
        yield "length-3-cut-off", bench_corpus._text(np.random.default_rng(1), 10 * n, dict_size=60000).tobytes()


TEXT_STREAMS = (("TEXT", "NONE"), ("TEXT", "ANS1"), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0"), ("TEXT+BWT+RANK+ZRLT", "ANS1"), ("TEXT+UTF", "HUFFMAN"),
                ("UTF+TEXT", "ANS0"), ("TEXT+TEXT", "FPAQ"), ("TEXT+UTF+BWT+SRT+ZRLT", "FPAQ"))      # the last one = the reference's -l 6 sequence


def check_text(be, n=200_000, chain=False, streams=TEXT_STREAMS, bs_stream=1 << 16):
    """(chain: the caller set KNZ_TEXT_CHAIN, the one-lane scan does the forward direction too.) TEXT transform objects (both stream formats: the entropy stage of the handle picks one, Factory.go:100-120) vs the oracle,
    then TEXT inside streams: text, UTF-8, binary and magic-number blocks side by side, ctx["dataType"] handed to the UTF stage."""
    import text_corpus as T
    for entropy, bs in (("ANS0", 4 << 20), ("ANS1", 4 << 20), ("ANS0", 1 << 16), ("ANS1", 1 << 16))[: 4 if n >= 200_000 else 3]:
        if True:
            c = K.Codec("NONE", entropy, bs, lib=be.lib)
            t = K.ByteTransform(c, "TEXT")
            applied = 0
            for name, data in text_inputs(n):
                O.set_ctx(bs, O.entropy_type(entropy))
                o = O.transform_forward(O.T_TEXT, data)
                g = t.forward(data)
                assert (g is None) == (o is None), (entropy, bs, name)
                if o is None:
                    continue
                assert g == o, (entropy, bs, name, len(g), len(o))
                assert c.last_counter(2) == (1 if chain else 0), (entropy, bs, name)       # the parallel kernel settled (or was not asked)
                applied += 1
                assert t.inverse(o, len(data) + 64) == data, (entropy, bs, name)
                assert c.last_counter(2) == (1 if chain else 0), (entropy, bs, name, "inverse")
            assert applied >= 11, applied
            c.close()
    if not streams:
        return
    bs = bs_stream
    rng = np.random.default_rng(9)
    data = (T.make_text(bs, seed=21) + T.make_text(bs, seed=22, crlf=True) + rng.integers(0, 256, bs).astype(np.uint8).tobytes() +
            T.make_text(bs, seed=23, utf8=0.3) + "".join(chr(0x410 + int(k)) for k in rng.integers(0, 60, bs // 2)).encode("utf-8") +
            bytes([0x1F, 0x8B]) + T.make_text(bs - 2, seed=24) + b"MZ" + T.make_text(bs - 2, seed=25) + T.make_text(bs // 3 + 5, seed=26, markup=True))
    for transform, entropy in streams:
        exp = O.compress(data, transform, entropy, bs)
        c = K.Codec(transform, entropy, bs, lib=be.lib)
        src, ks = be.to_dev(data)
        cap = 2 * len(data) + (1 << 20)
        dst, kd = be.empty(cap)
        nb = c.dev_compress(src, len(data), dst, cap)
        assert be.to_host(kd, nb) == exp, (transform, entropy)
        out, ko = be.empty(len(data) + 64)
        assert c.dev_decompress(dst, nb, out, len(data) + 64) == len(data)
        assert be.to_host(ko, len(data)) == data
        c.close()
    assert len(O.compress(data, "TEXT", "NONE", bs)) < len(data) - 3 * bs // 4
    # the data type a TEXT stage detects reaches an LZ stage behind it (LZCodec.go:298-311): DNA = minimum match 6, small alphabet = LZ declines
    dna = rng.choice(np.frombuffer(b"acgt", dtype=np.uint8), bs).tobytes()
    dna = dna[: bs // 2] + dna[: bs // 2]
    small = rng.choice(np.frombuffer(b"\x01\x02\x03", dtype=np.uint8), bs).tobytes()
    data2 = dna + small + T.make_text(bs // 2, seed=27)
    for transform, entropy in (("TEXT+LZ", "HUFFMAN"), ("TEXT+LZX", "ANS0")):
        exp = O.compress(data2, transform, entropy, bs)
        c = K.Codec(transform, entropy, bs, lib=be.lib)
        src, ks = be.to_dev(data2)
        cap = 2 * len(data2) + (1 << 20)
        dst, kd = be.empty(cap)
        nb = c.dev_compress(src, len(data2), dst, cap)
        assert be.to_host(kd, nb) == exp, (transform, entropy)
        out, ko = be.empty(len(data2) + 64)
        assert c.dev_decompress(dst, nb, out, len(data2) + 64) == len(data2)
        assert be.to_host(ko, len(data2)) == data2
        c.close()


def check_text_foreign_handle(be, n=60_000):
    """A stream is decoded with what ITS header says: TEXT streams (whose format follows ctx["entropy"] and whose hash size follows
    ctx["blockSize"], Factory.go:100-120, TextCodec.go:610-650) decoded through handles opened with another entropy codec and block size."""
    import text_corpus
    data = text_corpus.make_text(n, seed=11)
    for (tr, en, bs), (tr2, en2, bs2) in ((("TEXT", "ANS1", 1 << 15), ("NONE", "HUFFMAN", 1 << 20)), (("TEXT+UTF", "HUFFMAN", 1 << 14), ("TEXT", "FPAQ", 1 << 22)),
                                           (("TEXT", "NONE", 1 << 16), ("BWT", "ANS1", 1 << 13))):
        exp = O.compress(data, tr, en, bs)
        c = K.Codec(tr2, en2, bs2, lib=be.lib)
        sp, ks = be.to_dev(exp, 4)
        out, kout = be.empty(n + 4096)
        nd = c.dev_decompress(sp, len(exp), out, n + 4096)
        assert nd == n, (tr, en, bs, tr2, en2, bs2)
        assert be.to_host(kout, nd) == data, (tr, en, bs, tr2, en2, bs2)
        c.close()


def check_text_damaged(be, trials=60, n=20_000, seed=1):
    """TEXT inverse on damaged input: byte flips, truncations and spliced index bytes in valid encodings. Whatever the reference's scan does
    with them (fail, or decode to something else) both device paths must do as well: same bytes or an error where the oracle fails."""
    import text_corpus as T
    rng = np.random.default_rng(seed)
    for entropy in ("ANS0", "ANS1"):
        bs = 1 << 16
        c = K.Codec("NONE", entropy, bs, lib=be.lib)
        t = K.ByteTransform(c, "TEXT")
        base = T.make_text(n, seed=seed + 40, escapes=0.01, utf8=0.02, crlf=True)
        O.set_ctx(bs, O.entropy_type(entropy))
        good = O.transform_forward(O.T_TEXT, base)
        assert good is not None
        for k in range(trials):
            d = bytearray(good)
            kind = k % 6
            if kind == 5:                                                  # letters where delimiters were: the implied-space flag runs across letters
                for _ in range(int(rng.integers(1, 6))):
                    cand = [i for i in range(1, len(d)) if d[i] in b" _,.;-"]
                    d[cand[int(rng.integers(0, len(cand)))]] = ord("C")
            elif kind == 0:
                for _ in range(int(rng.integers(1, 4))):
                    d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
            elif kind == 1:
                d = d[: int(rng.integers(2, len(d)))]
            elif kind == 2:                                                # a run of index-like bytes dropped in
                p = int(rng.integers(1, len(d)))
                d[p:p] = bytes(int(v) for v in rng.choice([0x0F, 0x0E, 0x80, 0xC1, 0xF3, 0x41, 0x20, 0x7F, 0xFF], int(rng.integers(1, 6))))
            elif kind == 3:                                                # a reference to an entry far ahead of the dictionary
                p = int(rng.integers(1, len(d) - 4))
                d[p:p + 3] = bytes([0x0F, 0xE1, 0x85]) if entropy == "ANS1" else bytes([0xF1, 0x20, 0x33])
            else:
                p, q = sorted(int(v) for v in rng.integers(1, len(d), 2))
                d = d[:p] + d[q:]
            d = bytes(d)
            cap = len(base) + 4096
            O.set_ctx(bs, O.entropy_type(entropy))
            try:
                exp = O.transform_inverse(O.T_TEXT, d, cap)
            except O.OracleError:
                exp = None
            try:
                got = t.inverse(d, cap)
            except K.KnzError:
                got = None
            assert got == exp, (entropy, k, kind, None if exp is None else len(exp), None if got is None else len(got))
        c.close()


def check_concurrent_handles(be, threads=8, rounds=3):
    """the library is thread-safe for concurrent calls on different handles (one handle per goroutine / Writer).
    Several host threads, each with its own handle and its own configuration, compress and decompress at the same time."""
    import threading
    cfgs = [("NONE", "HUFFMAN", 1 << 16), ("NONE", "ANS0", 1 << 16), ("BWT+RANK+ZRLT", "ANS1", 1 << 15), ("LZ", "HUFFMAN", 1 << 16),
            ("NONE", "HUFFMAN", 4096), ("LZP", "ANS0", 1 << 16), ("ZRLT", "NONE", 1 << 14), ("BWT+SRT+ZRLT", "FPAQ", 1 << 14)]
    errors = []

    def work(i):
        try:
            transform, entropy, bs = cfgs[i % len(cfgs)]
            n = 200000 + 7777 * i
            data = corpus(n, 100 + i)
            exp = O.compress(data, transform, entropy, bs)
            c = K.Codec(transform, entropy, bs, lib=be.lib)
            src, ks = be.to_dev(data)
            cap = 2 * n + 262144 * (n // bs + 2)
            dst, kd = be.empty(cap)
            out, ko = be.empty(n + 64)
            for _ in range(rounds):
                nb = c.dev_compress(src, n, dst, cap)
                assert be.to_host(kd, nb) == exp, (i, "stream")
                assert c.dev_decompress(dst, nb, out, n + 64) == n
                assert be.to_host(ko, n) == data, (i, "round trip")
            c.close()
        except Exception as e:                       # noqa: BLE001 - reported below with the thread number
            errors.append((i, repr(e)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors


def check_deep_batches_several_handles(be, handles=4, depth=1024, bs=1 << 14, seed=3):
    """Several handles at EnableGPUDepth-sized ANS1 batches at the same time (round-5 verdict, weak #8): the fused ZRLT / RANK chain keeps two polling
    waves per block resident, and their number is bounded per DEVICE (KNZ_PIPE_DEVICE_BLOCKS): a handle that finds the budget spent takes the regular
    stage kernels. Every handle must get its blocks back, whichever path each batch took; together the handles may not have more than the budget piped."""
    import threading
    blocks = [corpus(bs, seed + (i % 37)) for i in range(depth)]
    enc = K.Codec("BWT+RANK+ZRLT", "ANS1", bs, lib=be.lib)
    pays = [r[0] for r in K.BlockBatch(enc).encode(blocks)]
    enc.close()
    codecs = [K.Codec("BWT+RANK+ZRLT", "ANS1", bs, lib=be.lib) for _ in range(handles)]
    errors, piped = [], [0] * handles

    def work(t):
        try:
            for _ in range(2):
                back = K.BlockBatch(codecs[t]).decode(pays)
                assert back == blocks, (t, "decoded blocks differ")
                piped[t] = codecs[t].last_counter(6)
        except Exception as e:                       # noqa: BLE001 - reported below with the thread number
            errors.append((t, repr(e)))

    ts = [threading.Thread(target=work, args=(t,)) for t in range(handles)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for c in codecs:
        c.close()
    assert not errors, errors
    return piped


def check_skip_blocks(be, light=False):
    """-s / ctx["skipBlocks"]: random blocks (entropy >= 973/1024) and blocks that start with a compressed-format magic number
    become copy blocks; the stream equals the oracle's for every entropy codec and decodes back."""
    r = np.random.default_rng(77)
    bs = 1 << 14 if light else 1 << 16                                # (the emulator runs the light form)
    text = corpus(bs, 5)
    rnd = r.integers(0, 256, bs, dtype=np.uint8).tobytes()
    gz = bytes([0x1F, 0x8B, 8, 0]) + corpus(bs - 4, 6)               # compressible, but says gzip
    jpg_e1 = bytes([0xFF, 0xD8, 0xFF, 0xE1]) + corpus(bs - 4, 7)      # recognised as JPG, not in the compressed list: stays
    almost = (r.integers(0, 256, bs, dtype=np.uint8) & 0x7F).tobytes()   # 7 bits per byte: under the threshold
    data = text + rnd + gz + text + jpg_e1 + almost + rnd[:12345]
    for transform, entropy in (("NONE", "HUFFMAN"), ("NONE", "ANS0"), ("NONE", "ANS1"), ("NONE", "FPAQ"), ("NONE", "NONE"),
                               ("BWT+RANK+ZRLT", "ANS0"), ("LZ", "HUFFMAN")):
        for ck in ((0,) if light and entropy != "HUFFMAN" else (0, 32)):
            exp = O.compress(data, transform, entropy, bs, ck, skip_blocks=True)
            assert exp != O.compress(data, transform, entropy, bs, ck) or (transform, entropy) == ("NONE", "NONE")
            c = K.Codec(transform, entropy, bs, checksum_bits=ck, lib=be.lib, skip_blocks=True)
            src, ks = be.to_dev(data)
            cap = 2 * len(data) + (1 << 20)
            dst, kd = be.empty(cap)
            nb = c.dev_compress(src, len(data), dst, cap)
            assert be.to_host(kd, nb) == exp, (transform, entropy, ck)
            out, ko = be.empty(len(data) + 64)
            assert c.dev_decompress(dst, nb, out, len(data) + 64) == len(data)
            assert be.to_host(ko, len(data)) == data
            c.close()


def _fuzz_data(r, n):
    kind = int(r.integers(0, 12))
    if kind >= 10:                                   # English-like text (TEXT territory): capitals, CR+LF, markup, escapes, UTF-8 letters
        import text_corpus as T
        return T.make_text(n, seed=int(r.integers(0, 1 << 30)), crlf=bool(r.integers(0, 2)), utf8=[0.0, 0.03][int(r.integers(0, 2))],
                           markup=bool(r.integers(0, 2)), escapes=[0.0, 0.01][int(r.integers(0, 2))], vocab=int(r.integers(50, 3000)))
    if kind >= 8:                                    # UTF-8 text (Cyrillic / CJK / emoji / ASCII words), sometimes cut inside a code point
        kinds = [(1,), (1, 1, 0), (2,), (3, 1), (0, 1, 2, 3)][int(r.integers(0, 5))]
        t = utf_text(n + 3, int(r.integers(0, 1 << 30)), kinds)
        o = int(r.integers(0, 3))
        return t[o:o + n]
    if kind == 0:
        return r.integers(0, 256, n, dtype=np.uint8).tobytes()
    if kind == 1:
        return corpus(n, int(r.integers(0, 1 << 30)))
    if kind == 2:                                    # runs
        out = bytearray()
        while len(out) < n:
            out += bytes([int(r.integers(0, 256))]) * int(r.integers(1, 200))
        return bytes(out[:n])
    if kind == 3:                                    # small alphabet, skewed
        return np.minimum(r.geometric(0.3, n), 255).astype(np.uint8).tobytes()
    if kind == 4:                                    # one symbol / two symbols
        v = r.integers(0, 256, 2, dtype=np.uint8)
        return (np.where(r.random(n) < 0.02, v[0], v[1])).astype(np.uint8).tobytes()
    if kind == 5:                                    # periodic records with a few edits (LZ / LZP territory)
        rec = r.integers(0, 256, int(r.integers(8, 300)), dtype=np.uint8)
        a = np.tile(rec, n // len(rec) + 1)[:n].copy()
        idx = r.integers(0, max(n, 1), max(n // 97, 1))
        a[idx] = r.integers(0, 256, len(idx), dtype=np.uint8)
        return a.tobytes()
    if kind == 6:                                    # mostly zeros with 0xFC / 0xFE / 0xFF sprinkled in (ZRLT / LZP escapes)
        a = np.zeros(n, dtype=np.uint8)
        idx = r.integers(0, max(n, 1), max(n // 13, 1))
        a[idx] = r.choice(np.array([0xFC, 0xFE, 0xFF, 1, 2, 0x80], dtype=np.uint8), len(idx))
        return a.tobytes()
    return (np.arange(n) * int(r.integers(1, 7)) >> int(r.integers(0, 5))).astype(np.uint8).tobytes()


def check_fuzz(be, cases, seed, max_n, heavy_max_n=None):
    """Seeded differential test: random transform sequences x entropy codecs x block sizes x checksum sizes x data shapes,
    device stream == oracle stream, the device decodes the oracle's stream, the oracle decodes the device's."""
    r = np.random.default_rng(seed)
    tnames = ["NONE", "BWT", "RANK", "MTFT", "ZRLT", "LZ", "LZX", "LZP", "SRT", "UTF", "TEXT"]
    enames = ["NONE", "HUFFMAN", "ANS0", "ANS1", "FPAQ"]
    heavy = {"BWT", "RANK", "MTFT", "SRT"}          # slow on the emulator (cross-lane heavy): smaller inputs there
    done = 0
    for case in range(cases):
        nt = int(r.integers(1, 4))
        seq = [tnames[int(i)] for i in r.integers(0, len(tnames), nt)]
        transform = "+".join(seq)
        entropy = enames[int(r.integers(0, len(enames)))]
        bs = int(r.choice([1024, 4096, 1 << 14, 1 << 16, 1 << 18]))
        lim = heavy_max_n if (heavy_max_n and heavy & set(seq)) else max_n
        n = int(r.integers(1, lim))
        if r.random() < 0.2:
            n = int(r.choice([1, 2, 15, 16, 31, 32, 33, bs - 1, bs, bs + 1, 2 * bs + 16]))
            n = max(1, min(n, lim))
        ck = int(r.choice([0, 0, 32, 64]))
        sk = bool(r.random() < 0.25)                 # -s: incompressible blocks become copy blocks
        data = _fuzz_data(r, n)
        tag = (case, transform, entropy, bs, n, ck, sk)
        try:
            exp = O.compress(data, transform, entropy, bs, ck, skip_blocks=sk)
        except O.OracleError as e:                   # inputs the reference itself fails on (Go panic => ERR_PROCESS_BLOCK)
            c = K.Codec(transform, entropy, bs, checksum_bits=ck, lib=be.lib, skip_blocks=sk)
            src, ks = be.to_dev(data)
            dst, kd = be.empty(2 * n + 262144 * (n // bs + 2))
            with pytest_raises_knz(e.code):
                c.dev_compress(src, n, dst, 2 * n + 262144 * (n // bs + 2))
            c.close()
            continue
        c = K.Codec(transform, entropy, bs, checksum_bits=ck, lib=be.lib, skip_blocks=sk)
        src, ks = be.to_dev(data)
        cap = 2 * n + 262144 * (n // bs + 2)
        dst, kd = be.empty(cap)
        nb = c.dev_compress(src, n, dst, cap)
        got = be.to_host(kd, nb)
        assert got == exp, tag
        sp, ksp = be.to_dev(exp, 4)
        out, ko = be.empty(n + 64)
        assert c.dev_decompress(sp, len(exp), out, n + 64) == n, tag
        assert be.to_host(ko, n) == data, tag
        if case % 5 == 0 and ck == 0 and not sk:     # the same input through the host-pointer batch hook (block-local bit strings)
            bb = K.BlockBatch(c)
            blocks = [data[i:i + bs] for i in range(0, n, bs)]
            res = bb.encode(blocks)
            tt, et = O.transform_type(transform), O.entropy_type(entropy)
            for blk, (bits, written, mode, post, skip) in zip(blocks, res):
                o = O.encode_block(blk, tt, et)
                assert (written, bits, mode, post, skip) == (o["written"], o["bits"], o["mode"], o["post_len"], o["skip_flags"]), tag
            assert bb.decode([x[0] for x in res]) == blocks, tag
        c.close()
        done += 1
    assert done >= cases * 3 // 4


def bwt_sort_inputs(scale=1):
    """Inputs that reach every part of the forward suffix sort (bwt_sort.hip): groups that stay on the segmented-sort list, groups larger than a
    segment (device-wide sort), groups that move from the large list to the normal one in several blocks of one batch (the list is then no longer in
    slot order), long runs, periodic data, blocks that end inside a repeat."""
    r = np.random.default_rng(4242)
    n = 9000 * scale
    z = np.zeros(n, dtype=np.uint8)
    idx = r.integers(0, n, n // 13)
    z[idx] = r.choice(np.array([0xFC, 0xFE, 0xFF, 1, 2, 0x80], dtype=np.uint8), len(idx))
    yield "sparse", z.tobytes()
    yield "zeros", bytes(n)
    yield "two", np.where(r.random(n) < 0.02, 7, 9).astype(np.uint8).tobytes()
    rec = r.integers(0, 256, 37, dtype=np.uint8)
    per = np.tile(rec, n // 37 + 1)[:n].copy()
    per[r.integers(0, n, n // 97)] = 3
    yield "periodic", per.tobytes()
    yield "text", corpus(n, 11)
    yield "runs", b"".join(bytes([int(r.integers(0, 4))]) * int(r.integers(1, 400)) for _ in range(n // 100))[:n]
    yield "abab", (b"ab" * n)[:n - 1]
    yield "small", np.minimum(r.geometric(0.5, n), 255).astype(np.uint8).tobytes()


def check_bwt_sort_forms(be, monkeypatch, scale=1, block_sizes=(1024, 4096), segs=("128", "256", None), check=True):
    """The forward suffix sort with small segments (emulator build: KNZ_EMU_SG_T, so that short inputs use the large list and the hand-over between
    the lists) and with the product's; the emulator build also checks the sort's invariants between rounds (KNZ_EMU_SS_CHECK)."""
    if check:
        monkeypatch.setenv("KNZ_EMU_SS_CHECK", "1")
    for seg in segs:
        if seg is None:
            monkeypatch.delenv("KNZ_EMU_SG_T", raising=False)
        else:
            monkeypatch.setenv("KNZ_EMU_SG_T", seg)
        for bs in block_sizes:
            c = K.Codec("BWT", "NONE", bs, lib=be.lib)
            for name, data in bwt_sort_inputs(scale):
                n = len(data)
                src, ks = be.to_dev(data)
                cap = 2 * n + 65536 + 64 * (n // bs + 2)
                dst, kd = be.empty(cap)
                nb = c.dev_compress(src, n, dst, cap)
                assert be.to_host(kd, nb) == O.compress(data, "BWT", "NONE", bs), (name, seg, bs)
            c.close()


def check_bwt_sort_fuzz(be, monkeypatch, cases=60, seed=7, max_n=40000, segs=("128", "256", "")):
    """Random shapes (runs, periodic data, sparse symbols, text, zeros) x block sizes x segment sizes through the forward suffix sort, stream == oracle;
    the emulator build checks the sort's invariants between rounds (700 such cases were run once when the sort was written: none failed)."""
    monkeypatch.setenv("KNZ_EMU_SS_CHECK", "1")
    r = np.random.default_rng(seed)
    for case in range(cases):
        T = str(r.choice(list(segs)))
        if T:
            monkeypatch.setenv("KNZ_EMU_SG_T", T)
        else:
            monkeypatch.delenv("KNZ_EMU_SG_T", raising=False)
        bs = int(r.choice([1024, 2048, 4096, 16384]))
        n = int(r.integers(1, max_n))
        kind = int(r.integers(0, 8))
        if kind == 0:
            data = _fuzz_data(r, n)
        elif kind == 1:
            data = bytes(n)
        elif kind == 2:
            data = np.where(r.random(n) < 0.01, 1, 0).astype(np.uint8).tobytes()
        elif kind == 3:
            per = r.integers(0, 3, int(r.integers(1, 9)), dtype=np.uint8)
            data = np.tile(per, n // len(per) + 1)[:n].tobytes()
        elif kind == 4:
            data = b"".join(bytes([int(r.integers(0, 3))]) * int(r.integers(1, 3000)) for _ in range(40))[:n] or b"x"
        elif kind == 5:
            data = corpus(n, int(r.integers(0, 1000)))
        elif kind == 6:
            data = np.minimum(r.geometric(0.6, n), 255).astype(np.uint8).tobytes()
        else:
            a = np.zeros(n, dtype=np.uint8)
            idx = r.integers(0, n, max(n // 50, 1))
            a[idx] = r.integers(1, 4, len(idx))
            data = a.tobytes()
        n = len(data)
        c = K.Codec("BWT", "NONE", bs, lib=be.lib)
        src, ks = be.to_dev(data)
        cap = 2 * n + 65536 + 64 * (n // bs + 2)
        dst, kd = be.empty(cap)
        nb = c.dev_compress(src, n, dst, cap)
        assert be.to_host(kd, nb) == O.compress(data, "BWT", "NONE", bs), (case, T, bs, n, kind)
        c.close()


def check_bwt_sort_wide_keys(be, monkeypatch, scale=1, block_sizes=(4096,), segs=("128", None), check=True):
    """Blocks of 64 MiB and more need more than 64 bits for (group id | run-aware key): the large list then keeps the plain doubling keys.
    KNZ_SS_RUN_BITS_PAD widens the run-aware payload for the decision only, so that small inputs with many large groups take that branch
    (pad 12: some rounds fit, some do not; pad 40: never fits); stream == oracle either way."""
    for pad in ("12", "40"):
        monkeypatch.setenv("KNZ_SS_RUN_BITS_PAD", pad)
        check_bwt_sort_forms(be, monkeypatch, scale=scale, block_sizes=block_sizes, segs=segs, check=check)
    monkeypatch.delenv("KNZ_SS_RUN_BITS_PAD", raising=False)


class pytest_raises_knz:
    def __init__(self, code):
        self.code = code

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        assert et is not None and issubclass(et, K.KnzError), "the device accepted an input the reference fails on"
        assert ev.code == self.code, (ev.code, self.code)
        return True


def check_corrupt_streams(be, trials=12):
    """Bit flips in valid streams: the device decoder must come back (an error code or some output), never hang or touch
    memory outside its buffers, for every codec on the path. When the oracle decodes the damaged stream without an error,
    the device must produce the same bytes (same decoder semantics), except where only the checksum could tell."""
    rng = np.random.default_rng(2024)
    n, bs = 150000, 65536
    data = corpus(n, 77)
    for transform, entropy in (("NONE", "HUFFMAN"), ("NONE", "ANS0"), ("NONE", "ANS1"), ("RANK+ZRLT", "ANS0"), ("LZ", "NONE"),
                               ("BWT", "NONE"), ("NONE", "FPAQ"), ("LZP", "NONE"), ("LZX", "NONE"), ("BWT+SRT+ZRLT", "NONE")):
        good = O.compress(data, transform, entropy, bs)
        c = K.Codec(transform, entropy, bs, lib=be.lib)
        out, ko = be.empty(n + bs + 64)
        for t in range(trials):
            bad = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                pos = int(rng.integers(24 * 8, len(bad) * 8))          # behind the stream header
                bad[pos >> 3] ^= 0x80 >> (pos & 7)
            sp, ks = be.to_dev(bytes(bad), 4)
            try:
                nd = c.dev_decompress(sp, len(bad), out, n + bs + 64)
                got = be.to_host(ko, nd)
            except K.KnzError as e:
                got = None
            try:
                exp = O.decompress(bytes(bad), n + bs + 64)
            except Exception:
                exp = None
            # a damaged ANS payload makes the reference read stale bytes of its own reusable buffer behind the payload
            # (v2/entropy/ANSRangeDecoder.go decodeChunk): those bytes are not part of the stream, so only the block checksum
            # can tell; for the other codecs the two decoders must agree on error-or-bytes
            # A damaged BWT block is a permutation with several cycles: the reference walks whatever cycle the primary index
            # is on and emits it without complaint, the device's chained inverse notices and reports ERR_PROCESS_BLOCK
            # (docs/HISTORY.md section 2, deviations).
            # (and an SRT block whose header frequencies no longer add up is reported by the device, walked by the reference)
            if entropy.startswith("ANS") or "BWT" in transform or "SRT" in transform:
                continue
            assert (exp is None) == (got is None), (transform, entropy, t, exp is None, got is None)
            if exp is not None:
                assert got == exp, (transform, entropy, t)
        c.close()
