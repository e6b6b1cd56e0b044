"""Reference-generated parity pin: .knz streams written by flanglet/kanzi-go's own Writer.

tests/golden/ref_streams/manifest.json holds, for 387 (input, configuration) cases, the length and sha256 of the stream the REFERENCE'S OWN CODE
writes (io/CompressedStream.go and everything under it), and the stream itself for the small inputs. The image has no Go toolchain: the
producer is oracle/_ref = the reference's .go files translated mechanically to C++ by tools/go2cpp (tools/make_ref_vectors_go2cpp.py, committed
recipe, runs in the build container); tools/make_ref_vectors.sh is the same recipe for a machine with Go and writes the same manifest.
These tests REQUIRE, for every case: oracle-encode == reference stream and oracle-decode(stream) == input (CPU suite), device-encode ==
reference stream and device-decode(stream) == input (GPU suite: neither /root/reference nor oracle/_ref is needed there, only the fixtures).
"""
import importlib.util
import json
import os
import warnings

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "tests", "golden", "ref_streams")
UNPINNED = ("PARITY UNPINNED: tests/golden/ref_streams/ holds no reference-generated streams. Run tools/make_ref_vectors_go2cpp.py (build container, "
            "needs /root/reference) or tools/make_ref_vectors.sh /path/to/kanzi-go (a machine with Go) and commit the result.")


def _gen():
    spec = importlib.util.spec_from_file_location("make_ref_inputs", os.path.join(ROOT, "tools", "make_ref_inputs.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _cases():
    """[(case dict, stream bytes or None)] for every manifest entry: the stream's sha256 is always there, its bytes for the small ones (or for all
    of them when the Go recipe wrote the files)."""
    mpath = os.path.join(REF_DIR, "manifest.json")
    if not os.path.exists(mpath):
        return []
    out = []
    for c in json.load(open(mpath))["cases"]:
        f = os.path.join(REF_DIR, c.get("file", c["name"] + ".knz"))
        stream = open(f, "rb").read() if os.path.exists(f) else None
        if stream is None and "sha256" not in c:
            continue
        out.append((c, stream))
    return out


def _same(got, c, ref):
    import hashlib
    if ref is not None:
        return got == ref
    return len(got) == c["stream_bytes"] and hashlib.sha256(got).hexdigest() == c["sha256"]


def _inputs():
    return dict(_gen().inputs())


def test_manifest_generator_is_deterministic(tmp_path):
    """The inputs the Go tool compresses are regenerated here from the same generators: same names, same bytes on every run."""
    g = _gen()
    a = {k: len(v) for k, v in g.inputs()}
    b = {k: len(v) for k, v in g.inputs()}
    assert a == b and len(a) >= 30
    g.main(str(tmp_path))
    m = json.load(open(tmp_path / "manifest.json"))
    assert len(m["cases"]) >= 100
    assert {c["entropy"] for c in m["cases"]} >= {"HUFFMAN", "ANS0", "ANS1", "FPAQ", "NONE"}
    committed = json.load(open(os.path.join(REF_DIR, "manifest.json")))
    assert [c["name"] for c in committed["cases"]] == [c["name"] for c in m["cases"]], "the committed manifest is not the generator's case list"


def test_committed_vectors_are_what_the_reference_writes_today():
    """Where /root/reference is present (build container): regenerate every stream with oracle/_ref and compare with the committed manifest."""
    import ref_lib as R
    if not R.can_build():
        pytest.skip("/root/reference is not here (GPU box): the committed fixtures stand on their own")
    data = _inputs()
    cases = _cases()
    assert len(cases) >= 300
    for c, ref in cases:
        src = data[c["input"][:-4]]
        for jobs in (1, 3):
            got = R.compress(src, c["transform"], c["entropy"], c["block_size"], c["checksum"], jobs=jobs, skip_blocks=c["skip_blocks"])
            assert _same(got, c, ref), (c["name"], jobs)
        assert R.decompress(got, len(src) + 64, jobs=2) == src, c["name"]


def test_oracle_against_reference_streams():
    cases = _cases()
    if not cases:
        warnings.warn(UNPINNED)
        pytest.skip(UNPINNED)
    data = _inputs()
    for c, ref in cases:
        src = data[c["input"][:-4]]
        assert len(src) == c["input_bytes"], c["name"]
        got = O.compress(src, c["transform"], c["entropy"], c["block_size"], c["checksum"], skip_blocks=c["skip_blocks"])
        assert _same(got, c, ref), f"{c['name']}: the oracle's stream differs from the reference's"
        assert O.decompress(got, len(src) + 64) == src, c["name"]          # (got IS the reference's stream at this point)


@pytest.mark.gpu
def test_device_against_reference_streams():
    cases = _cases()
    if not cases:
        warnings.warn(UNPINNED)
        pytest.skip(UNPINNED)
    import parity_cases as P
    import knz
    K = knz.package()
    be = P.GpuBackend()
    data = _inputs()
    for c, ref in cases:
        src = data[c["input"][:-4]]
        codec = K.Codec(c["transform"], c["entropy"], c["block_size"], c["checksum"], lib=be.lib, skip_blocks=c["skip_blocks"])
        n = len(src)
        sp, _k1 = be.to_dev(src)
        cap = 2 * n + 262144 * (n // c["block_size"] + 2)
        dst, kdst = be.empty(cap)
        nb = codec.dev_compress(sp, n, dst, cap)
        got = be.to_host(kdst, nb)
        assert _same(got, c, ref), f"{c['name']}: the device's stream differs from the reference's"
        rp, _k2 = be.to_dev(got, 4)                                              # (byte for byte the reference's stream)
        out, kout = be.empty(n + 4096)
        nd = codec.dev_decompress(rp, len(got), out, n + 4096)
        assert nd == n and be.to_host(kout, nd) == src, c["name"]
        codec.close()


def test_full_size_reference_vectors():
    """tests/golden/ref_streams/fullsize_manifest.json: the BASELINE configurations' full-size streams as the reference's Writer writes them (by sha256).
    Here (CPU suite): the manifest covers configs[1..3], the -l 5 preset and configs[4]'s block shape; the corpus generator still makes the input it
    was made from; the hand-written oracle writes the same stream for configs[1] (seconds); and, where /root/reference is present, oracle/_ref
    regenerates that entry bit for bit (the other entries take minutes of one core each: tools/make_ref_fullsize_vectors.py, and the GPU suite compares
    the device with every one of them)."""
    import hashlib
    import bench_corpus
    import ref_lib as R
    man = json.load(open(os.path.join(REF_DIR, "fullsize_manifest.json")))
    cases = {c["name"]: c for c in man["cases"]}
    assert {(c["transform"], c["entropy"], c["block_size"]) for c in man["cases"]} >= {("NONE", "HUFFMAN", 4 << 20), ("LZ", "ANS0", 4 << 20), ("BWT+RANK+ZRLT", "ANS1", 8 << 20),
                                                                                       ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0", 4 << 20), ("BWT+RANK+ZRLT", "FPAQ", 32 << 20)}
    c = cases["config1_huffman_4m"]
    data = bench_corpus.s_silesia().tobytes()
    assert len(data) == c["input_bytes"] and hashlib.sha256(data).hexdigest() == c["input_sha256"]
    got = O.compress(data, c["transform"], c["entropy"], c["block_size"], 0, jobs=os.cpu_count() or 1)
    assert len(got) == c["stream_bytes"] and hashlib.sha256(got).hexdigest() == c["sha256"], "oracle stream differs from the reference Writer's at full size"
    if R.can_build():
        ref = R.compress(data, c["transform"], c["entropy"], c["block_size"], 0, jobs=1)
        assert len(ref) == c["stream_bytes"] and hashlib.sha256(ref).hexdigest() == c["sha256"], "oracle/_ref no longer writes the committed full-size stream"
