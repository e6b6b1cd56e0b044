"""The oracle's TEXT transform (oracle/text.hpp, restating v2/transform/TextCodec.go): hand-derived encodings of small inputs for both
stream formats, round trips over synthetic text in every mode the codec distinguishes, data type detection for blocks that are not text."""
import numpy as np
import pytest

import oracle_lib as O
import text_corpus as T


def _fwd(data, entropy, block_size=4 << 20):
    O.set_ctx(block_size, entropy)
    return O.transform_forward(O.T_TEXT, data)


def _inv(data, n, entropy, block_size=4 << 20):
    O.set_ctx(block_size, entropy)
    return O.transform_inverse(O.T_TEXT, data, n)


def test_static_words_fixture():
    w = T.static_words()
    assert len(w) == 1024 and w[:7] == ["the", "be", "and", "of", "in", "to", "with"] and w[-1] == "united"


def test_static_word_references_codec1():
    # TextCodec.go:826-843: words found in the dictionary become 0x0F + index (varint, :936-953); a lone space between two references is dropped
    block = b"the be and of in to with " * 50
    out = _fwd(block, O.E_ANS1)
    exp = bytes([0]) + b"".join(bytes([0x0F, i]) for i in range(7)) * 50 + b" "
    assert out == exp
    assert _inv(out, len(block) + 16, O.E_ANS1) == block


def test_static_word_references_codec2():
    # :1380-1387, :1489-1511: index + 1 as 10xxxxxx ; 0x80 first = first letter's case flipped
    block = b"The be and of in to with " * 50
    out = _fwd(block, O.E_ANS0)
    exp = bytes([0]) + (bytes([0x80, 0x81]) + bytes(0x80 | (i + 1) for i in range(1, 7))) * 50 + b" "
    assert out == exp
    assert _inv(out, len(block) + 16, O.E_ANS0) == block


def test_codec_selection_follows_the_entropy_stage():
    # Factory.go:100-120
    block = b"the be and of in to with " * 50
    c2 = _fwd(block, O.E_HUFFMAN)
    assert c2 == _fwd(block, O.E_NONE) == _fwd(block, O.E_ANS0)
    assert _fwd(block, O.E_ANS1) == _fwd(block, O.E_FPAQ) == _fwd(block, None) != c2


def test_dynamic_words_and_escapes_codec1():
    # a 4-letter word enters the dictionary at index 1026 (1024 static + the two escape entries, :676-678) on its first occurrence and is
    # referenced afterwards: 0x0F, 0x80 | idx >> 7, idx & 0x7F ; a literal 0x0F is the reference to entry 1025
    block = (b"zqxj " * 4 + b"\x0f ") * 60
    out = _fwd(block, O.E_ANS1)
    ref = bytes([0x0F, 0x80 | (1026 >> 7), 1026 & 0x7F])
    esc = bytes([0x0F, 0x80 | (1025 >> 7), 1025 & 0x7F])
    exp = bytes([0]) + b"zqxj " + ref * 3 + b" " + esc + b" " + (ref * 4 + b" " + esc + b" ") * 59
    assert out == exp
    assert _inv(out, len(block) + 16, O.E_ANS1) == block


@pytest.mark.parametrize("entropy", [O.E_ANS0, O.E_ANS1])
@pytest.mark.parametrize("kw", [dict(), dict(crlf=True), dict(utf8=0.05), dict(markup=True), dict(escapes=0.01),
                                dict(vocab=50000, static_share=0.05), dict(upper=0.5), dict(max_word=40)])
def test_round_trip(entropy, kw):
    d = T.make_text(120_000, seed=11, **kw)
    out = _fwd(d, entropy)
    assert out is not None and len(out) < len(d)
    assert out[0] & 0x40 == (0x40 if kw.get("crlf") else 0)
    assert out[0] & 0x20 == (0x20 if kw.get("markup") else 0)
    assert _inv(out, len(d) + 64, entropy) == d
    assert O.data_type() == 1                                          # ctx["dataType"] = DT_TEXT


def test_not_text_sets_the_data_type():
    rng = np.random.default_rng(1)
    cases = {
        6: rng.choice(np.frombuffer(b"acgt", dtype=np.uint8), 5000).tobytes(),               # DT_DNA
        4: rng.choice(np.frombuffer(b"0123456789,. ", dtype=np.uint8), 5000).tobytes(),      # DT_NUMERIC
        5: rng.choice(np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/", dtype=np.uint8), 5000).tobytes(),  # DT_BASE64
        7: rng.integers(0, 256, 100000).astype(np.uint8).tobytes(),                          # DT_BIN: all 256 values present
        9: rng.choice(np.frombuffer(b"\x01\x02\x03", dtype=np.uint8), 5000).tobytes(),       # DT_SMALL_ALPHABET
        8: "".join(chr(0x410 + int(k)) for k in rng.integers(0, 60, 3000)).encode("utf-8"),         # DT_UTF8 (half the bytes are continuation bytes)
        0: bytes(range(1, 200)) * 30,                                                        # nothing recognisable
    }
    for dt, block in cases.items():
        # (letters only pass the strict statistics of codec 1 as text; codec 2 wants spaces, :246)
        assert _fwd(block, O.E_ANS0 if dt in (5, 6) else O.E_ANS1) is None
        assert O.data_type() == dt, dt


def test_small_and_magic_blocks_decline():
    assert _fwd(b"the be and " * 20, O.E_ANS1) is None                 # < 1024 bytes (:555-557)
    block = b"PK\x03\x04" + b"the be and of in to with " * 50          # a zip magic number: codec 2 (not strict) declines (:188-192)
    assert _fwd(block, O.E_ANS0) is None and O.data_type() == 0
    assert _fwd(block, O.E_ANS1) is not None                           # codec 1 is strict: counts the block anyway


def test_stream_with_text_stage():
    d = T.make_text(1 << 20, seed=5, utf8=0.02)
    for tn, en in (("TEXT+UTF+BWT+RANK+ZRLT", "ANS0"), ("TEXT+BWT+RANK+ZRLT", "ANS1"), ("TEXT", "NONE")):
        s = O.compress(d, O.transform_type(tn), O.entropy_type(en), 1 << 18, jobs=4)
        assert O.decompress(s, len(d) + 64, jobs=4) == d
