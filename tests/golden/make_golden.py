"""Generates tests/golden/*.json from constants held in the reference SOURCE (run in the build
container only; /root/reference does not exist on the GPU box). No reference code is copied: only
format constants that pin the oracle are extracted.

  python tests/golden/make_golden.py
"""
import json
import os
import re

REF = "/root/reference/v2"
HERE = os.path.dirname(os.path.abspath(__file__))


def expgolomb_tables():
    src = open(os.path.join(REF, "entropy/ExpGolombCodec.go")).read()
    body = src[src.index("_EXPG_VALUES"):]
    body = body[: body.index("// ExpGolombEncoder")]
    nums = [int(x) for x in re.findall(r"\b(\d+),", body)]
    assert len(nums) == 512, len(nums)
    return {"unsigned": nums[:256], "signed": nums[256:]}


def log2_4096_table():
    src = open(os.path.join(REF, "internal/Global.go")).read()
    body = src[src.index("LOG2_4096 = [...]uint32{"):]
    body = body[: body.index("}")]
    nums = [int(x) for x in re.findall(r"\b(\d+),", body)]
    assert len(nums) == 257, len(nums)
    return nums


def magic_values():
    src = open(os.path.join(REF, "internal/Magic.go")).read()
    return {m.group(1): int(m.group(2), 16) for m in re.finditer(r"(\w+_MAGIC\w*)\s*=\s*(0x[0-9A-Fa-f]+)", src)}


def text_codec_constants():
    """TextCodec.go:26-51 and the static dictionary (:97-186): groundwork for the TEXT stage (not built yet)."""
    src = open(os.path.join(REF, "transform/TextCodec.go")).read()
    body = src[src.index("_TC_DICT_EN_1024 = []byte(`") + len("_TC_DICT_EN_1024 = []byte(`"):]
    body = body[: body.index("`)")]
    words = "".join(ch for ch in body if ch.isalpha())            # createDictionary drops everything that is not a letter (:456-463)
    consts = {}
    for m in re.finditer(r"(_TC_[A-Z0-9_]+)\s*=\s*([^/\n]+)", src[: src.index("type dictEntry")]):
        expr = re.sub(r"\b(byte|int32)\(([^)]*)\)", r"\2", m.group(2).strip())       # byte(0x0F) -> 0x0F ; int32(-2073254261) -> -2073254261
        expr = re.sub(r"_TC_[A-Z0-9_]+", lambda k: str(consts[k.group(0)]), expr)
        consts[m.group(1)] = int(eval(expr, {"__builtins__": {}}))                  # plain integer arithmetic of the const block
    return {"static_dictionary_letters": words, "constants": consts}


def main():
    g = {
        "source": "flanglet/kanzi-go v2 (bitstream v6)",
        "expgolomb": expgolomb_tables(),
        # BWT.go:48-62 doc-comment example
        "bwt_mississippi": {"input": "mississippi", "bwt": "ipssmpissii", "primary_index": 5},
        # Entropy_test.go:54-67 values; expected length = 1 + number of 7-bit groups above the first
        "varint_values": [0, 1, 127, 128, 255, 16384, (1 << 21) - 1, 1 << 21, (1 << 28) - 1, 1 << 28, 0xFFFFFFFF],
        # CompressedStream.go:42-54
        # internal/Global.go:59-88 (4096*log2(x), x = 0..256) and entropy/EntropyUtils.go:26
        "log2_4096": log2_4096_table(), "incompressible_threshold": 973,
        # internal/Magic.go:22-58
        "magic": magic_values(),
        "text_codec": text_codec_constants(),
        "stream": {"magic": 0x4B414E5A, "version": 6, "hash_seed": 0x4B414E5A, "header_hash": 0x1E35A7BD,
                   "header_seed_mul": 0x01030507},
    }
    with open(os.path.join(HERE, "reference_constants.json"), "w") as f:
        json.dump(g, f, indent=1)
    print("wrote reference_constants.json")


if __name__ == "__main__":
    main()
