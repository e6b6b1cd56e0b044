#!/usr/bin/env python3
"""Inputs + manifest for tools/refgen (reference-generated .knz streams). Deterministic: bench_corpus.py generators (numpy PCG64) and
the 30 inputs extracted from the reference's own tests (tests/golden/reference_inputs.json). Writes <dir>/inputs/*.bin and
<dir>/manifest.json; the manifest (not the inputs) is what gets committed next to the streams."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench_corpus as bc  # noqa: E402

SLICE = 128 * 1024


def inputs():
    for kind, seed in (("text", 1), ("exe", 2), ("img16", 3), ("records", 4), ("db", 6), ("source", 8)):
        yield f"silesia_{kind}", bc._segment(kind, SLICE, seed).tobytes()
    yield "enwik", bc.s_enwik(SLICE).tobytes()
    yield "zeros", bytes(SLICE // 4)
    yield "random", np.random.Generator(np.random.PCG64(99)).integers(0, 256, SLICE // 4, dtype=np.uint8).tobytes()
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_inputs.json")))
    for i, item in enumerate(ref["entropy"] + ref["transform"]):
        data = bytes.fromhex(item["hex"])
        if len(data):
            yield f"reftest_{i:02d}", data


# (name, transform, entropy, block size, checksum bits, skip blocks): the five BASELINE.json configs as they are, the shipped presets they
# sit next to (-l 1, -l 5, -l 6; -l 2 without its DNA stage, which is outside SURVEY 8), -x 32 / 64, -s, and small-block variants
CONFIGS = [
    ("cfg1_huffman_1m", "NONE", "HUFFMAN", 1 << 20, 0, False), ("cfg2_huffman_4m", "NONE", "HUFFMAN", 4 << 20, 0, False),
    ("cfg3_lz_ans0_4m", "LZ", "ANS0", 4 << 20, 0, False), ("cfg4_bwt_ans1_8m", "BWT+RANK+ZRLT", "ANS1", 8 << 20, 0, False),
    ("cfg5_bwt_fpaq_32m", "BWT+RANK+ZRLT", "FPAQ", 32 << 20, 0, False),
    ("l1_lzx_none", "LZX", "NONE", 4 << 20, 0, False), ("l2nodna_lz_huffman", "LZ", "HUFFMAN", 4 << 20, 0, False),
    ("l5", "TEXT+UTF+BWT+RANK+ZRLT", "ANS0", 4 << 20, 0, False), ("l6", "TEXT+UTF+BWT+SRT+ZRLT", "FPAQ", 8 << 20, 0, False),
    ("x32_lz_ans0", "LZ", "ANS0", 1 << 16, 32, False), ("x64_bwt_ans1", "BWT+RANK+ZRLT", "ANS1", 1 << 16, 64, False),
    ("s_huffman", "NONE", "HUFFMAN", 1 << 16, 0, True),
    ("b64k_huffman", "NONE", "HUFFMAN", 1 << 16, 0, False), ("b64k_lz_ans0", "LZ", "ANS0", 1 << 16, 0, False),
    ("b64k_bwt_ans1", "BWT+RANK+ZRLT", "ANS1", 1 << 16, 0, False), ("b64k_bwt_fpaq", "BWT+RANK+ZRLT", "FPAQ", 1 << 16, 0, False),
    ("b64k_l5", "TEXT+UTF+BWT+RANK+ZRLT", "ANS0", 1 << 16, 0, False), ("lzp_huffman", "LZP", "HUFFMAN", 1 << 16, 0, False),
    ("mtft_ans0", "BWT+MTFT+ZRLT", "ANS0", 1 << 16, 0, False),
]


def main(out_dir):
    os.makedirs(os.path.join(out_dir, "inputs"), exist_ok=True)
    cases = []
    for iname, data in inputs():
        with open(os.path.join(out_dir, "inputs", iname + ".bin"), "wb") as f:
            f.write(data)
        small = iname.startswith("reftest_")
        for cname, tr, en, bs, ck, skip in CONFIGS:
            if small and not cname.startswith(("cfg", "x", "l5")):          # the reference's own test inputs: the BASELINE configs and a few more
                continue
            cases.append({"name": f"{cname}__{iname}", "input": iname + ".bin", "input_bytes": len(data), "transform": tr, "entropy": en,
                          "block_size": bs, "checksum": ck, "skip_blocks": skip})
    with open(os.path.join(out_dir, "manifest.json"), "w") as f:
        json.dump({"generator": "tools/make_ref_inputs.py", "cases": cases}, f, indent=0)
    print(f"{len(cases)} cases, inputs in {out_dir}/inputs")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ref_streams"))
