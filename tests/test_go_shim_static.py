"""Static checks of the cgo shim in go/ (no Go toolchain exists in the image, so the files cannot be compiled here): they parse with the Go parser of
tools/go2cpp, every C.knz_* function they call is declared in include/knz_gpu.h with that many parameters, every C.KNZ_* constant and C.knz_* type
they name exists there, and the three interface types implement the methods kanzi.ByteTransform / EntropyEncoder / EntropyDecoder ask for
(v2/Definitions.go:78-91,154-179)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "go2cpp"))
import goparse  # noqa: E402

GO_FILES = ["gpu_batch.go", "gpu_transform.go", "gpu_entropy.go", "testhooks/gpu_hooks_entropy_test.go", "testhooks/gpu_hooks_transform_test.go"]   # (the files that import "C")


def _walk(n, fn):
    if isinstance(n, goparse.Node):
        fn(n)
        for v in n.f.values():
            _walk(v, fn)
    elif isinstance(n, (list, tuple)):
        for v in n:
            _walk(v, fn)


def _header():
    text = open(os.path.join(ROOT, "include", "knz_gpu.h")).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(knz_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    consts = set(re.findall(r"\b(KNZ_\w+)\b", text))
    types = set(re.findall(r"\}\s*(knz_\w+)\s*;", text))
    return protos, consts, types


def test_go_shim_parses_and_matches_the_c_header():
    protos, consts, types = _header()
    assert {"knz_open", "knz_close", "knz_encode_blocks", "knz_decode_blocks", "knz_transform_forward", "knz_entropy_encode"} <= set(protos)
    calls, names = [], set()
    methods = {}
    for f in GO_FILES:
        ast = goparse.parse_file(os.path.join(ROOT, "go", f))
        assert any(path == "C" for _alias, path in ast.imports), f"{f}: no import \"C\""

        def visit(n):
            if n.kind == "Call" and n.fun.kind == "Selector" and n.fun.x.kind == "Ident" and n.fun.x.name == "C":
                calls.append((f, n.pos[1], n.fun.sel, len(n.args)))
            if n.kind == "Selector" and n.x.kind == "Ident" and n.x.name == "C":
                names.add(n.sel)
            if n.kind == "NamedType" and n.pkg == "C":
                names.add(n.name)
        _walk(ast.decls, visit)
        for d in ast.decls:
            if d.kind == "FuncDecl" and d.recv is not None:
                rt = d.recv.typ
                methods.setdefault(rt.elem.name if rt.kind == "PointerType" else rt.name, set()).add(d.name)
    assert len(calls) >= 10
    for f, line, name, nargs in calls:
        if name.startswith("knz_"):
            assert name in protos, f"go/{f}:{line}: C.{name} is not declared in include/knz_gpu.h"
            assert protos[name] == nargs, f"go/{f}:{line}: C.{name} takes {protos[name]} parameters, the shim passes {nargs}"
    for name in names:
        if name.startswith("KNZ_"):
            assert name in consts, f"C.{name} is not in include/knz_gpu.h"
        elif name.startswith("knz_") and name not in protos:
            assert name in types, f"C.{name} is neither a function nor a type of include/knz_gpu.h"
    # the Go interfaces of the boundary (v2/Definitions.go:78-91, 154-179)
    assert {"Forward", "Inverse", "MaxEncodedLen"} <= methods["GPUTransform"]
    enc = [t for t, ms in methods.items() if {"Write", "BitStream", "Dispose"} <= ms]
    dec = [t for t, ms in methods.items() if {"Read", "BitStream", "Dispose"} <= ms]
    assert enc and dec, methods


def test_every_go_file_of_the_shim_parses():
    """gpu_stream.go and the io test hook do not import "C" (they go through gpu_batch.go): they must still parse with the translator's Go parser"""
    for f in ("gpu_stream.go", "testhooks/gpu_hooks_io_test.go"):
        ast = goparse.parse_file(os.path.join(ROOT, "go", f))
        assert ast.package == "io" and len(ast.decls) >= 3, f


def test_library_exports_every_symbol_the_header_declares():
    """The C ABI is what the reference's FFI binds: every function include/knz_gpu.h declares is a dynamic symbol of the in-tree libknz_gpu.so (no compute
    call, no GPU needed: the symbol table is read with nm)."""
    import subprocess
    lib = os.path.join(ROOT, "kanzi-go_amd", "libknz_gpu.so")
    if not os.path.exists(lib):
        import pytest
        pytest.skip("libknz_gpu.so is not built (__graft_entry__.build())")
    protos, _consts, _types = _header()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.strip()}
    missing = sorted(set(protos) - exported)
    assert not missing, missing
    assert {"knz_open_devices", "knz_device_count", "knz_lane_count", "knz_last_lane_times"} <= set(protos)
