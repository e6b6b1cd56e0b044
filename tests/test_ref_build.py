"""oracle/_ref == the reference's own sources, compiled: kanzi-go's .go files translated mechanically to C++ by tools/go2cpp
(`make -C oracle _ref`), crossed here with the hand-written oracle (oracle/*.hpp) on every object of the hot path.

For each codec and each transform, on every input: _ref encode == oracle encode (bytes and bit count), _ref decode(oracle bits) ==
input, oracle decode(_ref bits) == input; declined inputs (Forward error = "skip me") must be declined by both. A hand restatement
and a mechanical translation of the same Go source agreeing on every case is the pin SURVEY 8c asks for when no Go toolchain exists:
a shared misreading of the reference by the restatement and by the device can no longer pass.
"""
import numpy as np
import pytest

import bench_corpus
import oracle_lib as O
import parity_cases as P
import ref_lib as R

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref is not built and /root/reference is not here to build it from")

ENTROPY = ("HUFFMAN", "ANS0", "ANS1", "FPAQ", "NONE")
TRANSFORMS = ("BWT", "RANK", "MTFT", "ZRLT", "LZ", "LZX", "LZP", "SRT", "TEXT", "UTF")


def corpus_slices(n=128 << 10):
    """the first n bytes of every generator of the bench corpus (S-silesia's six member kinds, S-enwik, S-rand, S-zero, S-ramp)"""
    for kind, seed in (("text", 1), ("exe", 2), ("img16", 3), ("records", 4), ("db", 6), ("source", 8)):
        yield f"silesia_{kind}", bench_corpus._segment(kind, n, seed).tobytes()
    yield "enwik", bench_corpus.s_enwik(n).tobytes()
    yield "rand", bench_corpus.s_rand(n).tobytes()
    yield "zero", bench_corpus.s_zero(n).tobytes()
    yield "ramp", bench_corpus.s_ramp(n).tobytes()


def reference_inputs():
    ent, trf = P.reference_test_inputs()
    return [(f"ref_{n}", d) for n, d in ent + trf if len(d) > 0]


def test_type_ids_come_from_the_reference():
    """the ids the tests and the product use are the reference's (transform/Factory.go GetType, entropy/EntropyCodecFactory.go GetType)"""
    for name in ENTROPY:
        assert R.entropy_type(name) == O.entropy_type(name), name
    for name in ("NONE", "BWT", "BWT+RANK+ZRLT", "TEXT+UTF+BWT+RANK+ZRLT", "LZ", "LZX", "LZP+SRT", "BWT+MTFT+ZRLT", "BWT+SRT+ZRLT"):
        assert R.transform_type(name) == O.transform_type(name), name


@pytest.mark.parametrize("ename", ENTROPY)
def test_entropy_codecs_ref_vs_oracle(ename):
    et = O.entropy_type(ename)
    cases = list(P.entropy_inputs()) + list(corpus_slices()) + reference_inputs()
    for name, data in cases:
        if ename == "ANS1" and len(data) in (2, 3):
            continue                                      # the reference panics there: test_ans1_short_chunk_panics
        rb, rbits = R.entropy_encode(et, data)
        ob, obits = O.entropy_encode(et, data)
        assert (rbits, rb) == (obits, ob), (ename, name, "_ref encode != oracle encode")
        d1, used1 = R.entropy_decode(et, ob, len(data))
        assert d1 == data and used1 == obits, (ename, name, "_ref decode(oracle bits)")
        d2, used2 = O.entropy_decode(et, rb, len(data))
        assert d2 == data and used2 == rbits, (ename, name, "oracle decode(_ref bits)")


def test_ans1_short_chunk_panics():
    """ANSRangeCodec.go:353-362 indexes block[-1] for an order-1 chunk of 2-3 bytes behind the <= 32-byte raw shortcut: the Go code panics
    (recovered as ERR_PROCESS_BLOCK by the block task); the translated code panics the same way and the oracle / the device mirror the error."""
    data = bytes(range(40)) * 104858                      # 4 MiB + 2 + ... -> last chunk of the order-1 coder holds 2 bytes
    data = data[: (4 << 20) + 2]
    with pytest.raises(R.RefError) as e:
        R.entropy_encode(O.E_ANS1, data)
    assert e.value.code == 3 and "index out of range" in str(e.value)
    with pytest.raises(Exception):
        O.entropy_encode(O.E_ANS1, data)


@pytest.mark.parametrize("tname", TRANSFORMS)
def test_transform_objects_ref_vs_oracle(tname):
    tid = P._TID[tname]
    cases = list(P.transform_inputs(zrlt=(tname == "ZRLT"))) + reference_inputs()
    if tname in ("UTF", "TEXT"):
        cases += list(P.utf_inputs()) + list(P.text_inputs(60000))
    if tname in ("LZ", "LZX", "LZP", "TEXT", "UTF", "RANK", "MTFT", "ZRLT", "SRT"):
        cases += list(corpus_slices(64 << 10))
    if tname in ("LZ", "LZX"):
        cases += [(f"lzf_{n}", d) for n, d in P.lz_forward_inputs()]
    done = declined = 0
    for name, data in cases:
        if len(data) == 0:
            continue
        for bs, ent in ((1 << 16, O.E_NONE), (4 << 20, O.E_ANS1)) if tname == "TEXT" else ((1 << 16, O.E_NONE),):
            O.set_ctx(bs, ent)
            R.set_ctx(bs, ent)
            o = O.transform_forward(tid, data)
            r = R.transform_forward(tid, data)
            assert (o is None) == (r is None), (tname, name, "one checker declines, the other does not")
            if o is None:
                declined += 1
                continue
            assert r == o, (tname, name, "_ref forward != oracle forward")
            cap = len(data) + 1024
            assert R.transform_inverse(tid, o, cap) == data, (tname, name, "_ref inverse(oracle forward)")
            assert O.transform_inverse(tid, r, cap) == data, (tname, name, "oracle inverse(_ref forward)")
            done += 1
    assert done >= 10, (tname, done, declined)


def test_bwt_object_and_known_answer():
    """BWT.go:48-62: "mississippi" -> "ipssmpissii", primary index 5, through the reference's own DivSufSort; then the primary indexes of
    8-chunk blocks and both inverse paths (inverseMergeTPSI below 4 MiB ... inverseBiPSIv2 above) against the oracle."""
    out, prim = R.bwt_forward(b"mississippi")
    assert out == b"ipssmpissii" and prim[0] == 5
    rng = np.random.default_rng(5)
    cases = [P.corpus(n, seed=n) for n in (255, 256, 257, 1000, 4099, 70000)] + [bytes(5000), bench_corpus._segment("exe", 300000, 2).tobytes()]
    cases.append(bench_corpus._segment("text", (4 << 20) + 4321, 1).tobytes())          # > 4 MiB: the biPSIv2 inverse
    for data in cases:
        ob, oprim = O.bwt_forward(data)
        rb, rprim = R.bwt_forward(data)
        chunks = 8 if len(data) >= 256 else 1
        assert rb == ob and rprim[:chunks] == oprim[:chunks], len(data)
        assert R.bwt_inverse(ob, oprim) == data, (len(data), "_ref inverse")
        assert O.bwt_inverse(rb, rprim) == data, (len(data), "oracle inverse")


@pytest.mark.parametrize("seq", ["BWT+RANK+ZRLT", "TEXT+UTF+BWT+RANK+ZRLT", "LZ", "LZX", "LZP+SRT", "BWT+MTFT+ZRLT", "BWT+SRT+ZRLT", "NONE"])
def test_sequences_and_skip_flags(seq):
    """transform.New(ctx, type): the ping-pong pipeline with its skip flags (Sequence.go:64-186), as the block task drives it"""
    tt = O.transform_type(seq)
    cases = [(n, d) for n, d in P.transform_inputs(max_len=100000)] + list(corpus_slices(96 << 10)) + list(P.utf_inputs())[:3]
    for name, data in cases:
        if len(data) == 0:
            continue
        O.set_ctx(1 << 20, O.E_ANS0)
        R.set_ctx(1 << 20, O.E_ANS0)
        O.lib().knzo_set_data_type(0)
        ob, oflags = O.sequence_forward(tt, data)
        rb, rflags = R.sequence_forward(tt, data)
        assert (rflags, rb) == (oflags, ob), (seq, name)
        cap = len(data) + 1024
        assert R.sequence_inverse(tt, oflags, ob, cap) == data, (seq, name, "_ref inverse")
        assert O.sequence_inverse(tt, rflags, rb, cap) == data, (seq, name, "oracle inverse")


def test_hashes_magic_and_entropy_estimate():
    L = O.lib()
    u8 = lambda b: np.frombuffer(b, dtype=np.uint8)
    for name, data in list(P.entropy_inputs()) + list(corpus_slices(20000)):
        a = u8(data)
        p = a.ctypes.data_as(O.C.POINTER(O.C.c_uint8))
        assert R.xxhash32(data, 0x4B414E5A) == L.knzo_xxhash32(p, len(a), 0x4B414E5A), name
        assert R.xxhash64(data, 0x4B414E5A) == L.knzo_xxhash64(p, len(a), 0x4B414E5A), name
        assert R.magic_type(data) == L.knzo_magic_type(p, len(a)), name
        if len(a):
            assert R.entropy1024(data) == L.knzo_entropy1024(p, len(a)), name
    # known answers of XXH32 (seed 0): the standard's test vectors
    assert R.xxhash32(b"", 0) == 0x02CC5D05
    assert R.xxhash32(b"a", 0) == 0x550D7456


def machine_code(arch, n=40000, seed=5):
    """bytes with the statistics the reference's EXE filter looks for (EXECodec.go detectExeType :738-809): binary, >= 10 % zeros, >= 1 % 0xFF,
    and relative calls / branches more often than one per 200 bytes -- x86 `E8 rel32` or ARM64 `BL imm26`"""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, n, dtype=np.uint8)
    a[rng.random(n) < 0.15] = 0
    a[rng.random(n) < 0.03] = 0xFF
    if arch == "x86":
        for i in range(16, n - 8, 48):
            rel = int(rng.integers(-30000, 30000))
            a[i] = 0xE8
            a[i + 1:i + 5] = np.frombuffer(int(rel & 0xFFFFFFFF).to_bytes(4, "little"), dtype=np.uint8)
    else:
        for i in range(16, n - 8, 32):
            word = 0x94000000 | int(rng.integers(0, 1 << 20))
            a[i:i + 4] = np.frombuffer(word.to_bytes(4, "little"), dtype=np.uint8)
    return a.tobytes()


@pytest.mark.parametrize("what", ["RANGE", "CM", "TPAQ", "TPAQX", "LZX", "LZP", "ROLZ", "ROLZX", "RLT", "SRT", "MM", "EXE", "BWTS", "PACK", "DNA"])
def test_rest_of_the_reference_round_trips(what):
    """The translated reference beyond the hot path (codecs that have no hand restatement to be crossed with): what its own encoder writes, its own
    decoder must read back. Bit-serial coders whose decoder calls into the predictor several times inside ONE expression (BinaryEntropyCodec.go
    DecodeByte) only work when the translation keeps Go's left-to-right order of calls: this is the case that found that rule missing."""
    inputs = [(n, d) for n, d in reference_inputs() if len(d) <= 1 << 16][:40] + [(n, d[:30000]) for n, d in corpus_slices(30000)]
    if what == "EXE":
        inputs = [("x86", machine_code("x86")), ("arm64", machine_code("arm64"))] + inputs[:10]
    if what == "DNA":
        inputs = [("acgt", bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.default_rng(3).integers(0, 4, 50000)]))] + inputs[:10]
    names = {"RANGE", "CM", "TPAQ", "TPAQX"}
    done = 0
    if what in names:
        et = R.entropy_type(what)
        for name, data in inputs:
            bits, nbits = R.entropy_encode(et, data)
            assert R.entropy_decode(et, bits, len(data))[0] == data, (what, name)
            done += 1
    else:
        tid = R.transform_type(what) >> 42                       # (one transform: the first 6-bit slot of the sequence id)
        for name, data in inputs:
            R.set_ctx(1 << 16, R.entropy_type("NONE"))
            f = R.transform_forward(tid, data)
            if f is None:
                continue                                          # declined (= the sequence skips it)
            assert R.transform_inverse(tid, f, len(data) + 1024) == data, (what, name)
            done += 1
    assert done >= (2 if what == "EXE" else 1), what
