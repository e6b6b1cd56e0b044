"""Regression vectors of the ORACLE (not of the reference: no Go toolchain exists in this image):
sha256 and length of the .knz stream the oracle writes for seeded inputs, one per transform / entropy / option combination on
the path. They pin the oracle's behaviour over time: a change that alters any stream has to be deliberate (rerun this script)
and cannot slip in together with a matching change on the device side.

  python tests/golden/make_oracle_vectors.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O          # noqa: E402
import parity_cases as P        # noqa: E402

CASES = [(t, e, bs, n, ck, sk)
         for t, e, bs, n in [("NONE", "NONE", 1024, 3000), ("NONE", "HUFFMAN", 1 << 16, 200000), ("NONE", "ANS0", 1 << 16, 200000),
                             ("NONE", "ANS1", 1 << 16, 200000), ("NONE", "FPAQ", 1 << 16, 100000), ("LZ", "ANS0", 1 << 16, 200000),
                             ("LZX", "NONE", 1 << 16, 150000), ("LZP", "HUFFMAN", 1 << 16, 200000), ("BWT", "NONE", 1 << 14, 50000),
                             ("BWT+RANK+ZRLT", "ANS1", 1 << 14, 50000), ("BWT+MTFT+ZRLT", "ANS0", 1 << 14, 50000),
                             ("BWT+SRT+ZRLT", "FPAQ", 1 << 14, 50000), ("RANK", "HUFFMAN", 4096, 20000), ("ZRLT", "NONE", 4096, 20000),
                             ("UTF", "HUFFMAN", 1 << 16, 150000), ("UTF+BWT+RANK+ZRLT", "ANS0", 1 << 15, 100000)]
         for ck, sk in [(0, False), (32, False), (64, True)]]


def data_for(t, n):
    return P.utf_text(n, 4242, (1, 2, 0)) if "UTF" in t else P.corpus(n, 4242)


def vectors():
    out = {}
    for t, e, bs, n, ck, sk in CASES:
        s = O.compress(data_for(t, n), t, e, bs, ck, skip_blocks=sk)
        out[f"{t}|{e}|{bs}|{n}|{ck}|{int(sk)}"] = {"len": len(s), "sha256": hashlib.sha256(s).hexdigest()}
    return out


if __name__ == "__main__":
    json.dump({"what": "oracle .knz streams for seeded inputs (parity_cases.corpus / utf_text, seed 4242)", "streams": vectors()},
              open(os.path.join(HERE, "oracle_streams.json"), "w"), indent=1)
    print("wrote oracle_streams.json")
