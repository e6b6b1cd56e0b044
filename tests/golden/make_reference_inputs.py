"""Extracts every deterministic input literal of the reference's own tests for the hot path from the reference SOURCE
(run in the build container only; /root/reference does not exist on the GPU box) into tests/golden/reference_inputs.json.
Nothing is re-typed: byte-slice literals, string literals and bytes.Repeat(...) expressions are parsed out of the cited line
ranges; the few loop-built inputs of those tests (all-identical, alternating, all 256 values, sparse) are described by a small
recipe whose parameters are read from the same lines.

  python tests/golden/make_reference_inputs.py

Sources: v2/entropy/Entropy_test.go:617-693, v2/transform/Transforms_test.go:165-258 and :534-601, v2/transform/BWT_test.go:60-84.
The reference's tests are round-trip tests (they hold no expected output bytes), so these are INPUTS: tests/parity_cases.py
replays them GPU-encode -> oracle-decode and oracle-encode -> GPU-decode.
"""
import json
import os
import re

REF = "/root/reference/v2"
HERE = os.path.dirname(os.path.abspath(__file__))


def go_bytes(expr):
    """[]byte{...} with ints / hex / 'c' runes, []byte("..."), bytes.Repeat(<bytes>, n)"""
    expr = expr.strip()
    m = re.match(r"bytes\.Repeat\((.*),\s*(\d+)\)$", expr, re.S)
    if m:
        return go_bytes(m.group(1)) * int(m.group(2))
    m = re.match(r'\[\]byte\("((?:[^"\\]|\\.)*)"\)$', expr)
    if m:
        return m.group(1).encode()
    m = re.match(r"\[\]byte\{(.*)\}$", expr, re.S)
    if m:
        out = bytearray()
        for tok in [t.strip() for t in m.group(1).split(",") if t.strip()]:
            out.append(ord(tok[1]) if tok.startswith("'") else int(tok, 0))
        return bytes(out)
    raise ValueError(expr)


def lines(path, lo, hi):
    return "".join(open(os.path.join(REF, path)).read().splitlines(keepends=True)[lo - 1:hi])


def entropy_inputs():
    path, lo, hi = "entropy/Entropy_test.go", 617, 693
    src = lines(path, lo, hi)
    out = []
    for m in re.finditer(r'\{name:\s*"([^"]+)",\s*input:\s*(\[\]byte\{[^}]*\}),\s*ii:\s*(\d+)\}', src):
        out.append({"name": m.group(1), "source": f"{path}:{lo}-{hi}", "hex": go_bytes(m.group(2)).hex()})
    # loop-built inputs: make([]byte, N) filled by byte(<expr of i>)
    for m in re.finditer(r'\{name:\s*"([^"]+)",\s*input:\s*func\(\)\s*\[\]byte\s*\{(.*?)\}\(\),\s*ii', src, re.S):
        name, body = m.group(1), m.group(2)
        n = int(re.search(r"make\(\[\]byte,\s*(\d+)\)", body).group(1))
        if "rand." in body:
            continue                                       # not deterministic in the reference either
        v = bytearray(n)
        if re.search(r"v\[i\]\s*=\s*byte\((\d+)\)", body):
            v = bytearray([int(re.search(r"v\[i\]\s*=\s*byte\((\d+)\)", body).group(1))]) * n
        elif "2 + (i & 1)" in body:
            v = bytearray(2 + (i & 1) for i in range(n))
        elif "v[i] = byte(i)" in body:
            v = bytearray(i & 255 for i in range(n))
        elif "byte('A')" in body and "byte('B')" in body:
            v = bytearray(ord("A") if i % 2 == 0 else ord("B") for i in range(n))
        elif "v[i*16] = byte(i)" in body:
            for i in range(1, 256):
                if i * 16 < n:
                    v[i * 16] = i
        else:
            raise ValueError(name)
        out.append({"name": name, "source": f"{path}:{lo}-{hi}", "hex": bytes(v).hex()})
    return out


def transform_inputs():
    path, lo, hi = "transform/Transforms_test.go", 165, 258
    src = lines(path, lo, hi)
    out = []
    for m in re.finditer(r'name:\s*"([^"]+)",\s*inputData:\s*(\[\]byte\{[^}]*\})', src):
        out.append({"name": m.group(1), "source": f"{path}:{lo}-{hi}", "hex": go_bytes(m.group(2)).hex()})
    assert "allByteValues[i] = byte(i)" in src
    out.append({"name": "All256ByteValues", "source": f"{path}:{lo}-{hi}", "hex": bytes(range(256)).hex()})
    m = re.search(r"input1 := make\(\[\]byte, (\d+)\).*?input1\[i\] = (\d+).*?input1\[0\] = (\d+)", src, re.S)
    v = bytearray([int(m.group(2))]) * int(m.group(1))
    v[0] = int(m.group(3))
    out.append({"name": "Original_AllEights_OneOne_80k_1", "source": f"{path}:{lo}-{hi}", "hex": bytes(v).hex()})
    path2, lo2, hi2 = "transform/Transforms_test.go", 534, 601
    src2 = lines(path2, lo2, hi2)
    for m in re.finditer(r'name:\s*"([^"]+)",\s*description:\s*"[^"]*",\s*input:\s*(.*?),\s*(?://[^\n]*)?\n', src2, re.S):
        name, expr = m.group(1), m.group(2).strip()
        if expr.startswith("func()"):
            if "rand." in expr:
                continue
            if "'A'" in expr and "'B'" in expr:
                n = int(re.search(r"make\(\[\]byte,\s*(\d+)\)", expr).group(1))
                out.append({"name": name, "source": f"{path2}:{lo2}-{hi2}", "hex": bytes(ord("A") if i % 2 == 0 else ord("B") for i in range(n)).hex()})
            continue
        if expr.startswith("append("):
            parts = re.findall(r'bytes\.Repeat\(\[\]byte\{[^}]*\},\s*\d+\)|\[\]byte\("[^"]*"\)', expr)
            # append(append(A, B...), C...) keeps source order
            data = b"".join(go_bytes(p) for p in parts)
        else:
            data = go_bytes(expr)
        out.append({"name": name, "source": f"{path2}:{lo2}-{hi2}", "hex": data.hex()})
    path3, lo3, hi3 = "transform/BWT_test.go", 60, 84
    src3 = lines(path3, lo3, hi3)
    for k, m in enumerate(re.finditer(r'buf1 = (\[\]byte\("[^"]*"\))', src3)):
        out.append({"name": f"BWT_test_string_{k + 1}", "source": f"{path3}:{lo3}-{hi3}", "hex": go_bytes(m.group(1)).hex()})
    return out


def main():
    g = {"source": "flanglet/kanzi-go v2 test sources (inputs only: the reference's tests check round trips, they hold no output bytes)",
         "entropy": entropy_inputs(), "transform": transform_inputs()}
    assert len(g["entropy"]) >= 10 and len(g["transform"]) >= 14, (len(g["entropy"]), len(g["transform"]))
    with open(os.path.join(HERE, "reference_inputs.json"), "w") as f:
        json.dump(g, f, indent=1)
    print("entropy:", [(e["name"], len(e["hex"]) // 2) for e in g["entropy"]])
    print("transform:", [(e["name"], len(e["hex"]) // 2) for e in g["transform"]])


if __name__ == "__main__":
    main()
