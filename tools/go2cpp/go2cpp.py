#!/usr/bin/env python3
"""go2cpp: mechanical translation of kanzi-go source files to one C++20 translation unit (oracle/_ref).

TEST INFRASTRUCTURE. usage: go2cpp.py --root /root/reference/v2 --out oracle/_ref/kanzi_ref.gen.hpp FILE.go ...
FILEs are relative to --root (or PKGDIR=/path/to/file.go for a file that lives elsewhere: the cgo shim of go/, a patched copy); files are
grouped into packages by directory, packages are emitted in the order their first file is listed (list dependencies first). The generated file is never edited and never committed.
"""
import argparse
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import goparse  # noqa: E402
import emit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--root", default=".")
    ap.add_argument("--module", default="", help="module path (default: read from ROOT/go.mod)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--cgo", action="store_true", help="the file set holds cgo files: include the cgo shim (tools/go2cpp/runtime/cgo_shim.hpp)")
    ap.add_argument("files", nargs="+")
    a = ap.parse_args()
    if a.module:
        module = a.module
    else:
        with open(os.path.join(a.root, "go.mod")) as f:
            module = re.search(r"^module\s+(\S+)", f.read(), re.M).group(1)
    tr = emit.Translator(module)
    for rel in a.files:
        if "=" in rel:                                   # PKGDIR=FILE: a file from somewhere else (the cgo shim, a patched copy) that belongs to package directory PKGDIR
            d, path = rel.split("=", 1)
            d = d.strip("/")
            if d == ".":
                d = ""
        else:
            d, path = os.path.dirname(rel), os.path.join(a.root, rel)
        ast = goparse.parse_file(path)
        tr.add_file(module + ("/" + d if d else ""), ast)
    text = tr.emit_all()
    if a.cgo:
        text = text.replace('#include "go_rt.hpp"', '#include "go_rt.hpp"\n#include "cgo_shim.hpp"', 1)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        f.write(text)
    print(f"go2cpp: {len(a.files)} files, {sum(len(p.funcs) + sum(len(m) for m in p.methods.values()) for p in tr.pkgs.values())} functions -> {a.out} ({len(text.splitlines())} lines)")


if __name__ == "__main__":
    try:
        main()
    except (goparse.GoSyntaxError, emit.EmitError) as e:
        sys.exit(f"go2cpp: {e}")
