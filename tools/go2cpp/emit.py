"""C++20 emitter for the Go subset parsed by goparse.py.

TEST INFRASTRUCTURE (oracle/_ref). The emitter carries Go's semantics mechanically; it knows nothing about compression:
  * every sized integer is a go::I<T> (wrap-around arithmetic, Go shift rules), untyped constants are go::U (128 bits) and take
    the type of their context exactly where Go gives them one; the C++ type checker then plays the role of Go's
    (mixing go::I<int32_t> with go::I<int64_t> does not compile, as in Go);
  * Go's operator precedence is kept by full parenthesisation;
  * slices / arrays / strings are bounds-checked views that panic (C++ exception) where Go panics;
  * structs get `operator->` so that `x.f` is emitted as `x->f` for values, pointers and interfaces alike;
  * interfaces are abstract classes, satisfied structurally: a struct inherits every known interface whose method set it has;
  * multiple results are std::tuple, `a, b := f()` unpacks, `switch` never falls through unless asked, `break` / `continue`
    with labels become gotos, `go f(x)` runs f(x) in place (the one goroutine fan-out of BWT.go is serialised).
Anything outside the subset raises: the translator never guesses.
"""
import os
import sys

from goparse import Node


class EmitError(Exception):
    pass


BUILTIN_TYPES = {
    "int": "go::Int", "uint": "go::Uint", "uintptr": "go::Uintptr", "int8": "go::Int8", "int16": "go::Int16", "int32": "go::Int32",
    "int64": "go::Int64", "uint8": "go::Uint8", "uint16": "go::Uint16", "uint32": "go::Uint32", "uint64": "go::Uint64", "byte": "go::Byte",
    "rune": "go::Rune", "float64": "go::Float64", "float32": "go::Float32", "bool": "bool", "string": "go::String", "error": "go::error",
    "any": "go::any",
}
INT_RAW = {"int": "int64_t", "uint": "uint64_t", "int8": "int8_t", "int16": "int16_t", "int32": "int32_t", "int64": "int64_t", "uint8": "uint8_t",
           "uint16": "uint16_t", "uint32": "uint32_t", "uint64": "uint64_t", "byte": "uint8_t", "rune": "int32_t", "uintptr": "uint64_t"}
CPP_KEYWORDS = {"new", "delete", "class", "template", "register", "default", "int", "long", "short", "char", "signed", "unsigned", "auto", "union",
                "typename", "namespace", "operator", "private", "public", "protected", "virtual", "friend", "inline", "static", "float", "double",
                "void", "bool", "true_", "try", "catch", "throw", "using", "typedef", "enum", "extern", "volatile", "mutable", "explicit", "export",
                "do", "while", "and", "or", "not", "xor", "asm", "concept", "requires", "constexpr", "const_cast", "sizeof", "typeid", "alignas",
                "alignof", "decltype", "noexcept", "nullptr", "static_assert", "thread_local", "wchar_t", "bitand", "bitor", "compl", "errno",
                "stdin", "stdout", "stderr", "NULL", "EOF", "I", "U", "main", "signal", "unix", "linux"}
# interfaces of the standard library the translated files use (method name -> (param types, result types) as C++ strings)
RW_SIG = (("go::Slice<go::Byte>",), ("go::Int", "go::error"))
STD_INTERFACES = {
    ("io", "Reader"): {"Read": RW_SIG},
    ("io", "Writer"): {"Write": RW_SIG},
    ("io", "Closer"): {"Close": ((), ("go::error",))},
    ("io", "ReadCloser"): {"Read": RW_SIG, "Close": ((), ("go::error",))},
    ("io", "WriteCloser"): {"Write": RW_SIG, "Close": ((), ("go::error",))},
    ("io", "ReadWriteCloser"): {"Read": RW_SIG, "Write": RW_SIG, "Close": ((), ("go::error",))},
    ("go", "error_iface"): {"Error": ((), ("go::String",))},
}
STD_VALUE_TYPES = {("sync", "WaitGroup"): "::go_sync::WaitGroup", ("sync", "Mutex"): "::go_sync::Mutex", ("bytes", "Buffer"): "::go_bytes::Buffer", ("time", "Time"): "::go_time::Time", ("unsafe", "Pointer"): "::go_unsafe::Pointer", ("runtime", "Pinner"): "::go_runtime::Pinner",
                   ("testing", "T"): "::go_testing::T", ("testing", "B"): "::go_testing::T", ("rand", "Rand"): "::go_rand::Rand", ("bytes", "Reader"): "::go_bytes::Reader", ("os", "File"): "::go_os::File"}
# cgo: the C types the shim files name (include/knz_gpu.h and <stdint.h> through tools/go2cpp/runtime/cgo_shim.hpp)
CGO_TYPES = {"uint8_t", "uint16_t", "uint32_t", "uint64_t", "int8_t", "int16_t", "int32_t", "int64_t", "int", "uint", "size_t", "char", "float", "knz_cfg", "knz_block"}
STD_PACKAGES = {"errors", "fmt", "encoding/binary", "math/bits", "sort", "slices", "io", "sync", "strings", "bytes", "sync/atomic", "time", "runtime", "C", "unsafe", "testing", "math/rand", "os"}


THIS_IS_RECEIVER = [False]     # Go's conventional receiver name `this` is C++'s `this` inside methods; elsewhere it is an ordinary local


def mangle(name):
    if name == "this":
        return "this" if THIS_IS_RECEIVER[0] else "this_"
    return name + "_" if name in CPP_KEYWORDS else name


def cstr(b):
    out = []
    for c in b:
        if c == 34:
            out.append('\\"')
        elif c == 92:
            out.append("\\\\")
        elif 32 <= c < 127 and c != 63:
            out.append(chr(c))
        else:
            out.append("\\%03o" % c)
    return '"' + "".join(out) + '"'


def walk_idents(n, out):
    """names referenced by an expression / type tree (for the dependency order of package-level consts and vars)"""
    if isinstance(n, Node):
        if n.kind == "Ident":
            out.add(n.name)
        elif n.kind == "NamedType" and n.pkg is None:
            out.add(n.name)
        for v in n.f.values():
            walk_idents(v, out)
    elif isinstance(n, (list, tuple)):
        for v in n:
            walk_idents(v, out)


def base_ident(x):
    """the variable an addressable expression lives in: a.b[i].c -> a (None when it starts from a call, a deref ...)"""
    while True:
        if x.kind == "Ident":
            return x.name
        if x.kind == "Selector":
            x = x.x
        elif x.kind == "Index":
            x = x.x
        elif x.kind == "Paren":
            x = x.x
        else:
            return None


def escaping_names(n, out):
    """Go moves a local to the heap when its address may outlive the frame. Without Go's escape analysis every local whose address is taken
    (&x, &x.f, &x[i]) or that is sliced (x[a:b] of an array) is given heap (arena) storage: conservative, never wrong."""
    if isinstance(n, Node):
        if n.kind == "Unary" and n.op == "&" and n.x.kind != "Composite":
            b = base_ident(n.x)
            if b is not None:
                out.add(b)
        elif n.kind == "SliceExpr":
            b = base_ident(n.x)
            if b is not None:
                out.add(b)
        elif n.kind == "Defer":
            walk_idents(n.call, out)           # a deferred closure runs after the body's block is gone: everything it names gets heap storage
        for v in n.f.values():
            escaping_names(v, out)
    elif isinstance(n, (list, tuple)):
        for v in n:
            escaping_names(v, out)


def closures_naming(n, names, out):
    """function literals inside n whose bodies name one of `names` (loop variables: Go >= 1.22 gives every iteration its own copy, a C++ reference
    capture does not: refused rather than translated wrongly)"""
    if isinstance(n, Node):
        if n.kind == "FuncLit":
            ids = set()
            walk_idents(n.body, ids)
            declared = {p.name for p in n.sig.params if p.name}
            if (ids - declared) & names:
                out.append(n)
            return
        for v in n.f.values():
            closures_naming(v, names, out)
    elif isinstance(n, (list, tuple)):
        for v in n:
            closures_naming(v, names, out)


PURE_BUILTINS = ("len", "cap", "min", "max", "new", "make")


def has_call(tr, n):
    """does evaluating expression n run a function or method call (conversions and the side-effect-free builtins do not count; a function literal's
    body is not evaluated where it stands)"""
    if isinstance(n, Node):
        if n.kind == "FuncLit":
            return False
        if n.kind == "Call":
            if tr.resolve_type_expr(n.fun) is not None:
                return has_call(tr, n.args)
            f = n.fun
            if f.kind == "Ident" and not tr.is_local(f.name) and f.name not in tr.pk.funcs and f.name in PURE_BUILTINS:
                return has_call(tr, n.args)
            return True
        return any(has_call(tr, v) for v in n.f.values())
    if isinstance(n, (list, tuple)):
        return any(has_call(tr, v) for v in n)
    return False


def has_defer(n):
    if isinstance(n, Node):
        if n.kind == "Defer":
            return True
        if n.kind == "FuncLit":
            return False
        return any(has_defer(v) for v in n.f.values())
    if isinstance(n, (list, tuple)):
        return any(has_defer(v) for v in n)
    return False


class Package:
    def __init__(self, name, path):
        self.name, self.path = name, path
        self.ns = "kz_" + name
        self.files = []
        self.types = {}      # name -> TypeSpec
        self.funcs = {}      # name -> FuncDecl
        self.methods = {}    # type name -> [FuncDecl]
        self.consts = {}     # name -> (ConstSpec, index)
        self.vars = {}       # name -> (VarSpec, index)
        self.inits = []


class Translator:
    def __init__(self, module_path):
        self.module_path = module_path      # e.g. github.com/flanglet/kanzi-go/v2
        self.pkgs = {}                      # import path -> Package
        self.order = []
        self.tmp = 0
        self.ltypes = [{}]
        self.hoisted = []
        self.hoisted_names = {}
        self.iface_sigs = {}                # C++ interface type -> {method: (params, results)}
        for (p, n), sig in STD_INTERFACES.items():
            self.iface_sigs[("::go_%s::%s" % (p, n)) if p != "go" else "::go::error_iface"] = sig

    # ------------------------------------------------------------------------------------------------ loading
    def add_file(self, import_path, ast):
        pk = self.pkgs.get(import_path)
        if pk is None:
            pk = Package(ast.package, import_path)
            self.pkgs[import_path] = pk
            self.order.append(import_path)
        if pk.name != ast.package:
            raise EmitError(f"{ast.pos[0]}: package {ast.package} in {import_path} (expected {pk.name})")
        pk.files.append(ast)
        for d in ast.decls:
            if d.kind == "TypeSpec":
                pk.types[d.name] = d
            elif d.kind == "FuncDecl":
                if d.recv is not None:
                    rt = d.recv.typ
                    tn = rt.elem.name if rt.kind == "PointerType" else rt.name
                    pk.methods.setdefault(tn, []).append(d)
                elif d.name == "init":
                    pk.inits.append(d)
                else:
                    pk.funcs[d.name] = d
            elif d.kind == "ConstSpec":
                for i, nm in enumerate(d.names):
                    pk.consts[nm] = (d, i)
            elif d.kind == "VarSpec":
                for i, nm in enumerate(d.names):
                    pk.vars[nm] = (d, i)

    def tmpname(self, stem="t"):
        self.tmp += 1
        return f"_{stem}{self.tmp}"

    def err(self, node, msg):
        pos = node.pos if isinstance(node, Node) and node.pos else ("?", 0)
        raise EmitError(f"{pos[0]}:{pos[1]}: {msg}")

    # ------------------------------------------------------------------------------------------------ per-file context
    def set_file(self, pk, ast):
        self.pk = pk
        self.imports = {}
        for alias, path in ast.imports:
            self.imports[alias] = path

    def pkg_of_alias(self, alias):
        """-> ('kz', Package) | ('std', name) | None"""
        path = self.imports.get(alias)
        if path is None:
            return None
        if path in self.pkgs:
            return ("kz", self.pkgs[path])
        if path in STD_PACKAGES:
            return ("std", path.split("/")[-1])
        raise EmitError(f"import {path!r} is neither a translated package nor a shimmed standard package")

    def ns_of_alias(self, alias):
        r = self.pkg_of_alias(alias)
        return ("::" + r[1].ns) if r[0] == "kz" else ("::go_" + r[1])

    # ------------------------------------------------------------------------------------------------ scopes
    def push(self):
        self.scopes.append(set())
        self.ltypes.append({})

    def pop(self):
        self.scopes.pop()
        self.ltypes.pop()

    def local_type(self, name):
        """(hoisted C++ name, type node) of a type declared inside the current function, innermost scope first"""
        for sc in reversed(getattr(self, "ltypes", [])):
            if name in sc:
                return sc[name]
        return None

    def hoist_struct(self, t, stem):
        """a struct type that has no package-level name (a local `type X struct`, an anonymous `struct{...}`): defined at namespace scope just before the
        function that uses it"""
        key = id(t)
        if key in self.hoisted_names:
            return self.hoisted_names[key]
        self.tmp += 1
        cn = f"{stem}_{self.tmp}"
        self.hoisted_names[key] = cn
        lines = [f"struct {cn} {{", f"  {cn}* operator->() {{ return this; }}", f"  const {cn}* operator->() const {{ return this; }}"]
        for f in t.fields:
            if f.name != "_":
                lines.append(f"  {self.ctype(f.typ)} {mangle(f.name)}{{}};")
        lines.append("};")
        self.hoisted.append("\n".join(lines))
        return cn

    def declare(self, name):
        self.scopes[-1].add(name)

    def is_local(self, name):
        return any(name in s for s in self.scopes)

    # ------------------------------------------------------------------------------------------------ types
    def type_kind(self, pk, name):
        ts = pk.types.get(name)
        if ts is None:
            return None
        k = ts.typ.kind
        return {"StructType": "struct", "InterfaceType": "interface"}.get(k, "named")

    def ctype(self, t):
        k = t.kind
        if k == "StructType":
            return self.hoist_struct(t, "Anon")
        if k == "NamedType":
            if t.pkg is None:
                lt = self.local_type(t.name)
                if lt is not None:
                    return lt[0]
                if t.name in self.pk.types and not self.is_local_type_shadow(t.name):
                    return self.named_ctype(self.pk, t.name)
                if t.name in BUILTIN_TYPES:
                    return BUILTIN_TYPES[t.name]
                self.err(t, f"unknown type {t.name}")
            r = self.pkg_of_alias(t.pkg)
            if r is None:
                self.err(t, f"unknown package {t.pkg}")
            if r[0] == "kz":
                if t.name not in r[1].types:
                    self.err(t, f"unknown type {t.pkg}.{t.name}")
                return self.named_ctype(r[1], t.name)
            if r[1] == "C":
                if t.name not in CGO_TYPES:
                    self.err(t, f"C type {t.name} is not in the cgo shim")
                return f"::go_C::{mangle(t.name)}"
            if (r[1], t.name) in STD_INTERFACES:
                return f"::go_{r[1]}::{t.name}*"
            if (r[1], t.name) in STD_VALUE_TYPES:
                return STD_VALUE_TYPES[(r[1], t.name)]
            self.err(t, f"standard type {t.pkg}.{t.name} has no shim")
        if k == "PointerType":
            return self.ctype(t.elem) + "*"
        if k == "SliceType":
            return f"go::Slice<{self.ctype(t.elem)}>"
        if k == "ArrayType":
            if t.len is None:
                self.err(t, "[...]T outside a composite literal")
            return f"go::Array<{self.ctype(t.elem)}, go::csize({self.ex(t.len)})>"
        if k == "MapType":
            return f"go::Map<{self.ctype(t.key)}, {self.ctype(t.elem)}>"
        if k == "FuncType":
            ps = ", ".join(self.ctype(p.typ) for p in t.sig.params)
            return f"std::function<{self.result_ctype(t.sig)}({ps})>"
        if k == "InterfaceType" and not t.methods and not t.embeds:
            return "go::any"
        self.err(t, f"type {k} is not supported here")

    def is_local_type_shadow(self, name):
        return False

    def named_ctype(self, pk, name):
        kind = self.type_kind(pk, name)
        base = f"::{pk.ns}::{mangle(name)}"
        return base + "*" if kind == "interface" else base

    def result_ctype(self, sig):
        rs = sig.results
        if not rs:
            return "void"
        if len(rs) == 1:
            return self.ctype(rs[0].typ)
        return "std::tuple<" + ", ".join(self.ctype(r.typ) for r in rs) + ">"

    def param_ctype(self, p):
        t = self.ctype(p.typ)
        return f"go::Slice<{t}>" if p.variadic else t

    def params_decl(self, sig, names=True):
        out = []
        for i, p in enumerate(sig.params):
            nm = mangle(p.name) if (p.name and p.name != "_") else f"_unused{i}"
            if p.name in getattr(self, "heap_params", ()):
                nm += "_arg"
            out.append(self.param_ctype(p) + (" " + nm if names else ""))
        return ", ".join(out)

    # ------------------------------------------------------------------------------------------------ expressions
    def resolve_type_expr(self, x):
        """If expression node x denotes a type, return its AST type node, else None."""
        if x.kind == "TypeExpr":
            return x.typ
        if x.kind == "Paren":
            return self.resolve_type_expr(x.x)
        if x.kind == "Ident":
            if self.local_type(x.name) is not None and not self.is_local(x.name):
                return Node("NamedType", x.pos, pkg=None, name=x.name)
            if self.is_local(x.name):
                return None
            if x.name in self.pk.types or (x.name in BUILTIN_TYPES and x.name not in self.pk.funcs):
                return Node("NamedType", x.pos, pkg=None, name=x.name)
            return None
        if x.kind == "Selector" and x.x.kind == "Ident" and not self.is_local(x.x.name):
            r = self.pkg_of_alias(x.x.name)
            if r is None:
                return None
            if r[0] == "kz" and x.sel in r[1].types:
                return Node("NamedType", x.pos, pkg=x.x.name, name=x.sel)
            if r[0] == "std" and ((r[1], x.sel) in STD_INTERFACES or (r[1], x.sel) in STD_VALUE_TYPES or (r[1] == "C" and x.sel in CGO_TYPES)):
                return Node("NamedType", x.pos, pkg=x.x.name, name=x.sel)
            return None
        if x.kind == "Unary" and x.op == "*":
            inner = self.resolve_type_expr(x.x)
            if inner is not None:
                return Node("PointerType", x.pos, elem=inner)
        return None

    def ex(self, x):
        k = x.kind
        m = getattr(self, "ex_" + k, None)
        if m is None:
            self.err(x, f"expression {k} is not supported")
        return m(x)

    def ex_IntLit(self, x):
        v = x.value
        if v >= 1 << 64:
            self.err(x, "integer literal beyond 64 bits")
        return f"{v}_u" if v < 1 << 63 else f"0x{v:X}_u"

    def ex_CharLit(self, x):
        return f"{x.value}_u"

    def ex_FloatLit(self, x):
        return repr(float(x.value))

    def ex_StringLit(self, x):
        return f"go::String({cstr(x.value)}, {len(x.value)})"

    def ex_Paren(self, x):
        return "(" + self.ex(x.x) + ")"

    def ex_Ident(self, x):
        n = x.name
        if self.is_local(n):
            return mangle(n)
        if n == "iota":
            if self.iota is None:
                self.err(x, "iota outside a const declaration")
            return f"{self.iota}_u"
        if n in ("true", "false"):
            return n
        if n == "nil":
            return "go::nil"
        if n in self.pk.consts or n in self.pk.vars or n in self.pk.funcs:
            return f"::{self.pk.ns}::{mangle(n)}"
        if n == "_":
            self.err(x, "blank identifier used as a value")
        self.err(x, f"unresolved identifier {n}")

    def ex_Unary(self, x):
        op = x.op
        if op == "&":
            inner = x.x
            if inner.kind == "Composite":
                return f"go::New({self.ex(inner)})"
            return f"(&({self.ex(inner)}))"
        if op == "*":
            return f"(*({self.ex(x.x)}))"
        cop = {"-": "-", "+": "+", "!": "!", "^": "~"}[op]
        return f"({cop}({self.ex(x.x)}))"

    def ordered(self, operands, strs, build, node=None):
        """Go runs the calls among the operands of one expression in lexical left-to-right order; C++ leaves the order of the operands of most
        operators and of the arguments of a call open. Where two or more operands run calls, each such operand is evaluated into a temporary, in
        order, before the expression itself. operands: AST nodes (None = not an expression), strs: their C++ text, build: strs -> C++ text."""
        calling = [i for i, o in enumerate(operands) if o is not None and has_call(self, o)]
        if len(calling) < 2:
            return build(strs)
        if os.environ.get("GO2CPP_REPORT_ORDER") and node is not None and node.pos:
            sys.stderr.write(f"ordered: {node.pos[0]}:{node.pos[1]} ({len(calling)} operands with calls)\n")
        pre, out = [], list(strs)
        for i in calling:
            t = self.tmpname("o")
            pre.append(f"auto {t} = {strs[i]};")
            out[i] = t
        cap = "&" if getattr(self, "in_func", False) else ""
        return f"([{cap}]{{ {' '.join(pre)} return {build(out)}; }}())"

    def ex_Binary(self, x):
        a, b = self.ex(x.x), self.ex(x.y)
        if x.op in ("&&", "||"):
            return f"({a} {x.op} {b})"
        if x.op == "&^":
            return self.ordered([x.x, x.y], [a, b], lambda v: f"andnot({v[0]}, {v[1]})", x)
        return self.ordered([x.x, x.y], [a, b], lambda v: f"({v[0]} {x.op} {v[1]})", x)

    def ex_Selector(self, x):
        if x.x.kind == "Ident" and not self.is_local(x.x.name) and x.x.name in self.imports:
            r = self.pkg_of_alias(x.x.name)
            if r[0] == "kz":
                pk = r[1]
                if x.sel in pk.consts or x.sel in pk.vars or x.sel in pk.funcs:
                    return f"::{pk.ns}::{mangle(x.sel)}"
                self.err(x, f"{x.x.name}.{x.sel} is not a const, var or func of a translated package")
            nm = x.sel
            if r[1] == "io" and nm == "EOF":
                nm = "EOF_"
            return f"::go_{r[1]}::{nm}"
        return f"{self.ex(x.x)}->{mangle(x.sel)}"

    def ex_Index(self, x):
        return f"{self.ex(x.x)}[{self.ex(x.index)}]"

    def ex_SliceExpr(self, x):
        lo = self.ex(x.lo) if x.lo is not None else "go::none"
        hi = self.ex(x.hi) if x.hi is not None else "go::none"
        if x.three:
            return self.ordered([x.x, x.lo, x.hi, x.max], [self.ex(x.x), lo, hi, self.ex(x.max)], lambda v: f"go::slice3({', '.join(v)})", x)
        return self.ordered([x.x, x.lo, x.hi], [self.ex(x.x), lo, hi], lambda v: f"go::slice({', '.join(v)})", x)

    def ex_TypeAssert(self, x):
        return f"go::assert1<{self.ctype(x.typ)}>({self.ex(x.x)})"

    def ex_TypeExpr(self, x):
        self.err(x, "type used as a value")

    def ex_FuncLit(self, x):
        sig = x.sig
        self.push()
        saved = (self.cur_results, self.named_results)
        self.cur_results = sig.results
        self.named_results = [r.name for r in sig.results] if sig.results and sig.results[0].name else None
        for p in sig.params:
            if p.name:
                self.declare(p.name)
        self.in_closure += 1
        body = self.func_body(x.body, sig)
        self.in_closure -= 1
        self.cur_results, self.named_results = saved
        self.pop()
        return f"[&]({self.params_decl(sig)}) -> {self.result_ctype(sig)} {body}"

    def find_callee(self, fun):
        """FuncDecl of a call to a package-level function of a translated package (for variadic packing), else None"""
        if fun.kind == "Ident" and not self.is_local(fun.name):
            return self.pk.funcs.get(fun.name)
        if fun.kind == "Selector" and fun.x.kind == "Ident" and not self.is_local(fun.x.name) and fun.x.name in self.imports:
            r = self.pkg_of_alias(fun.x.name)
            if r[0] == "kz":
                return r[1].funcs.get(fun.sel)
        return None

    def ex_Call(self, x):
        fun = x.fun
        # conversion?
        t = self.resolve_type_expr(fun)
        if t is not None:
            if len(x.args) != 1:
                self.err(x, "conversion with != 1 argument")
            return f"go::conv<{self.ctype(t)}>({self.ex(x.args[0])})"
        if fun.kind == "Ident" and not self.is_local(fun.name) and fun.name not in self.pk.funcs:
            b = self.builtin_call(x)
            if b is not None:
                return b
        callee = self.find_callee(fun)
        fn = self.ex(fun)

        def build(args):
            if callee is not None and callee.sig.params and callee.sig.params[-1].variadic and not x.spread:
                nfix = len(callee.sig.params) - 1
                et = self.ctype_in_pkg(callee)
                rest = args[nfix:]
                packed = f"go::Slice<{et}>::lit({{{', '.join(rest)}}})" if rest else f"go::Slice<{et}>()"
                args = args[:nfix] + [packed]
            return f"{fn}({', '.join(args)})"
        # (the function value / receiver expression is sequenced before the arguments by C++17; the arguments among themselves are not)
        return self.ordered(list(x.args), [self.ex(a) for a in x.args], build, x)

    def ctype_in_pkg(self, callee):
        """element type of the variadic parameter of `callee` (types resolved in the callee's own package / imports)"""
        for pk in self.pkgs.values():
            for f in pk.files:
                if callee in f.decls:
                    saved = (self.pk, self.imports)
                    self.set_file(pk, f)
                    try:
                        return self.ctype(callee.sig.params[-1].typ)
                    finally:
                        self.pk, self.imports = saved
        raise EmitError("callee not found")

    def builtin_call(self, x):
        n = x.fun.name
        a = x.args
        if n in ("len", "cap"):
            return f"go::{n}({self.ex(a[0])})"
        if n == "copy":
            return self.ordered(a[:2], [self.ex(a[0]), self.ex(a[1])], lambda v: f"go::copy({v[0]}, {v[1]})", x)
        if n in ("min", "max"):
            return self.ordered(list(a), [self.ex(v) for v in a], lambda v: f"go::{n}({', '.join(v)})", x)
        if n == "clear":
            return f"go::clear({self.ex(a[0])})"
        if n == "panic":
            return f"go::panic({self.ex(a[0])})"
        if n == "append":
            if x.spread:
                if len(a) != 2:
                    self.err(x, "append with spread and several values")
                return self.ordered(a[:2], [self.ex(a[0]), self.ex(a[1])], lambda v: f"go::append_slice({v[0]}, {v[1]})", x)
            return self.ordered(list(a), [self.ex(v) for v in a], lambda v: f"go::append({', '.join(v)})", x)
        if n == "make":
            t = self.resolve_type_expr(a[0])
            if t is None:
                self.err(x, "make of a non-type")
            if t.kind == "SliceType":
                et = self.ctype(t.elem)
                return self.ordered(list(a[1:]), [self.ex(v) for v in a[1:]], lambda v: f"go::make_slice<{et}>({', '.join(v)})", x)
            if t.kind == "MapType":
                return f"go::make_map<{self.ctype(t.key)}, {self.ctype(t.elem)}>()"
            self.err(x, "make of this type is not supported")
        if n == "new":
            t = self.resolve_type_expr(a[0])
            return f"go::New<{self.ctype(t)}>()"
        if n == "recover":
            return "go::recover()"
        if n == "delete":
            return self.ordered(a[:2], [self.ex(a[0]), self.ex(a[1])], lambda v: f"go::map_delete({v[0]}, {v[1]})", x)
        if n == "println":
            return f"go_fmt::Println({', '.join(self.ex(v) for v in a)})"
        if n in ("print", "complex", "real", "imag", "close"):
            self.err(x, f"builtin {n} is not supported")
        return None

    # composite literals
    def ex_Composite(self, x, expected=None):
        if x.typ is None:
            if expected is None:
                self.err(x, "composite literal without a type")
            t = expected
        else:
            t = self.resolve_type_expr(x.typ)
            if t is None:
                self.err(x, "composite literal of a non-type")
        return self.composite(x, t)

    def elem_value(self, v, elem_type):
        if v.kind == "Composite" and v.typ is None:
            return self.composite(v, elem_type)
        if v.kind == "Unary" and v.op == "&" and v.x.kind == "Composite" and v.x.typ is None:
            return f"go::New({self.composite(v.x, elem_type.elem)})"
        return self.ex(v)

    def composite(self, x, t):
        k = t.kind
        if k == "SliceType":
            if any(e.key is not None for e in x.elems):
                self.err(x, "keyed slice literal")
            et = self.ctype(t.elem)
            return f"go::Slice<{et}>::lit({{{', '.join(self.typed_elem(e.value, t.elem, et) for e in x.elems)}}})"
        if k == "ArrayType":
            if any(e.key is not None for e in x.elems):
                self.err(x, "keyed array literal")
            et = self.ctype(t.elem)
            n = f"go::csize({self.ex(t.len)})" if t.len is not None else str(len(x.elems))
            if not x.elems:
                return f"go::Array<{et}, {n}>{{}}"
            return f"go::Array<{et}, {n}>{{{{{', '.join(self.typed_elem(e.value, t.elem, et) for e in x.elems)}}}}}"
        if k == "NamedType" or k == "StructType":
            lt = self.local_type(t.name) if (k == "NamedType" and t.pkg is None) else None
            if k == "StructType":
                fields = t.fields
            elif lt is not None:
                fields = lt[1].fields
            else:
                pk = self.pk if t.pkg is None else self.pkg_of_alias(t.pkg)[1]
                if isinstance(pk, str):
                    if not x.elems:
                        return f"{self.ctype(t)}{{}}"
                    self.err(x, "composite literal of a standard type")
                ts = pk.types.get(t.name)
                if ts is None or ts.typ.kind != "StructType":
                    if ts is not None and ts.typ.kind in ("SliceType", "ArrayType"):
                        return self.composite(x, ts.typ)
                    self.err(x, f"composite literal of non-struct {t.name}")
                fields = ts.typ.fields
            ct = self.ctype(t)
            if not x.elems:
                return f"{ct}{{}}"
            v = self.tmpname("v")
            parts = []
            saved = (self.pk, self.imports)
            for i, e in enumerate(x.elems):
                if e.key is not None:
                    if e.key.kind != "Ident":
                        self.err(x, "struct literal key")
                    fname = e.key.name
                    ftyp = next((f.typ for f in fields if f.name == fname), None)
                else:
                    fname, ftyp = fields[i].name, fields[i].typ
                parts.append(f"{v}.{mangle(fname)} = {self.elem_value(e.value, ftyp)};")
            return f"[&]{{ {ct} {v}{{}}; {' '.join(parts)} return {v}; }}()"
        if k == "MapType":
            mk = f"go::make_map<{self.ctype(t.key)}, {self.ctype(t.elem)}>()"
            if not x.elems:
                return mk
            v = self.tmpname("m")
            to_any = t.elem.kind == "InterfaceType"
            parts = []
            for e in x.elems:
                if e.key is None:
                    self.err(x, "map literal element without a key")
                val = self.elem_value(e.value, t.elem)
                parts.append(f"{v}[{self.elem_value(e.key, t.key)}] = {('go::def(' + val + ')') if to_any else val};")
            cap = "&" if getattr(self, "in_func", False) else ""
            return f"[{cap}]{{ auto {v} = {mk}; {' '.join(parts)} return {v}; }}()"
        self.err(x, f"composite literal of {k}")

    def typed_elem(self, v, elem_type, et):
        s = self.elem_value(v, elem_type)
        return s

    # ------------------------------------------------------------------------------------------------ statements
    def block(self, b, new_scope=True):
        if new_scope:
            self.push()
        lines = ["{"]
        for s in b.stmts:
            lines.append(self.stmt(s))
        lines.append("}")
        if new_scope:
            self.pop()
        return "\n".join(lines)

    def stmt(self, s):
        m = getattr(self, "st_" + s.kind, None)
        if m is None:
            self.err(s, f"statement {s.kind} is not supported")
        return m(s)

    def st_Block(self, s):
        return self.block(s)

    def st_ExprStmt(self, s):
        return self.ex(s.x) + ";"

    def st_IncDec(self, s):
        return f"{self.ex(s.x)}{s.op};"

    def st_Go(self, s):
        return f"{self.ex(s.call)};   // (go statement: run in place)"

    def st_Defer(self, s):
        if not self.defer_frame:
            self.err(s, "defer inside a closure is not supported")
        c = s.call
        if c.kind != "Call":
            self.err(s, "defer of a non-call")
        if c.fun.kind == "FuncLit" and not c.args:
            return f"_df.push({self.ex(c.fun)});"
        # defer f(args): the arguments are evaluated now, the call runs at exit
        tmps, pre = [], []
        for a in c.args:
            t = self.tmpname("da")
            pre.append(f"auto& {t} = *go::New(go::def({self.ex(a)}));")
            tmps.append(t)
        fn = self.ex(c.fun)
        return "\n".join(pre) + f"\n_df.push([&]() {{ {fn}({', '.join(tmps)}); }});"

    def st_Labeled(self, s):
        inner = s.stmt
        lab = mangle(s.label)
        if inner is not None and inner.kind in ("For", "RangeFor", "Switch"):
            self.loop_label = s.label
            body = self.stmt(inner)
            return f"{lab}:;\n{body}\n{lab}_break:;"
        return f"{lab}:;\n" + (self.stmt(inner) if inner is not None else "")

    def st_Branch(self, s):
        if s.tok == "goto":
            return f"goto {mangle(s.label)};"
        if s.tok == "fallthrough":
            if self.fall_var is None:
                self.err(s, "fallthrough outside a switch")
            return f"{self.fall_var[0]} = {self.fall_var[1]}; goto {self.fall_var[2]};"
        if s.label is None:
            return f"{s.tok};"
        if s.tok == "break":
            return f"goto {mangle(s.label)}_break;"
        self.used_continue_labels.add(s.label)
        return f"goto {mangle(s.label)}_continue;"

    def st_Return(self, s):
        if self.defer_frame and self.in_closure == 0:
            return self.defer_return(s)
        return self.plain_return(s)

    def defer_return(self, s):
        """inside a function that defers: results go to the frame's result variables, then control leaves the try block"""
        rs = self.cur_results
        if not rs:
            return "goto _df_done;"
        names = [mangle(n) for n in self.named_results] if self.named_results else [f"_ret{i}" for i in range(len(rs))]
        if not s.values:
            return "goto _df_done;"
        if len(s.values) == len(rs):
            tmps = []
            out = ["{"]
            for v in s.values:
                t = self.tmpname()
                tmps.append(t)
                out.append(f"auto {t} = {self.ex(v)};")
            out += [f"{n} = {t};" for n, t in zip(names, tmps)]
            out.append("goto _df_done; }")
            return " ".join(out)
        if len(s.values) == 1:
            t = self.tmpname()
            return f"{{ auto {t} = {self.ex(s.values[0])}; " + " ".join(f"{n} = std::get<{i}>({t});" for i, n in enumerate(names)) + " goto _df_done; }"
        self.err(s, "return arity")

    def plain_return(self, s):
        rs = self.cur_results
        if not s.values:
            if not rs:
                return "return;"
            if self.named_results is None:
                self.err(s, "bare return without named results")
            if len(rs) == 1:
                return f"return {mangle(self.named_results[0])};"
            return f"return {self.result_tuple()}({', '.join(mangle(n) for n in self.named_results)});"
        if len(rs) == 1:
            if len(s.values) != 1:
                self.err(s, "return arity")
            return f"return {self.ex(s.values[0])};"
        if len(s.values) == 1:
            return f"return {self.ex(s.values[0])};"          # return f() with the same result list
        if len(s.values) != len(rs):
            self.err(s, "return arity")
        rt = self.result_tuple()
        return "return " + self.ordered(list(s.values), [self.ex(v) for v in s.values], lambda v: f"{rt}({', '.join(v)})", s) + ";"

    def result_tuple(self):
        return "std::tuple<" + ", ".join(self.ctype(r.typ) for r in self.cur_results) + ">"

    def st_DeclStmt(self, s):
        out = []
        for spec in s.specs:
            if spec.kind == "VarSpec":
                out.append(self.local_var(spec))
            elif spec.kind == "ConstSpec":
                self.iota = spec.iota
                for i, nm in enumerate(spec.names):
                    val = self.ex(spec.values[i])
                    if nm == "_":
                        continue
                    if spec.typ is not None:
                        out.append(f"constexpr {self.ctype(spec.typ)} {mangle(nm)} = {val};")
                    else:
                        out.append(f"constexpr auto {mangle(nm)} = {val};")
                    self.declare(nm)
                self.iota = None
            elif spec.kind == "TypeSpec" and spec.typ.kind == "StructType":
                self.ltypes[-1][spec.name] = (self.hoist_struct(spec.typ, "L_" + mangle(spec.name)), spec.typ)
            else:
                self.err(s, "local type declarations other than structs are not supported")
        return "\n".join(out)

    def local_var(self, spec):
        out = []
        if spec.values is None:
            ct = self.ctype(spec.typ)
            for nm in spec.names:
                self.declare(nm)
                if self.heap(nm):
                    out.append(f"{ct}& {mangle(nm)} = *go::New<{ct}>();")
                else:
                    out.append(f"{ct} {mangle(nm)}{{}};")
            return "\n".join(out)
        if len(spec.values) == len(spec.names):
            vals = [self.value_for(v, spec.typ) for v in spec.values]
            for nm, v in zip(spec.names, vals):
                if nm == "_":
                    out.append(f"(void)({v});")
                    continue
                if spec.typ is not None:
                    ct = self.ctype(spec.typ)
                    out.append(f"{ct}& {mangle(nm)} = *go::New<{ct}>({ct}({v}));" if self.heap(nm) else f"{ct} {mangle(nm)} = {v};")
                else:
                    t = self.tmpname()
                    out.append(f"auto {t} = go::def({v});")           # (the initialiser may name an outer variable of the same name)
                    out.append(f"auto& {mangle(nm)} = *go::New({t});" if self.heap(nm) else f"auto {mangle(nm)} = {t};")
            for nm in spec.names:
                self.declare(nm)
            return "\n".join(out)
        if len(spec.values) == 1:
            return self.unpack(spec.names, spec.values[0], define=True, all_new=True)
        self.err(spec, "var arity")

    def value_for(self, v, typ):
        if v.kind == "Composite" and v.typ is None and typ is not None:
            return self.composite(v, typ)
        return self.ex(v)

    def multi_value(self, rhs, n):
        """C++ expression of a tuple for a Go expression that yields n values (call, comma-ok map index, comma-ok type assertion)"""
        if rhs.kind == "Call":
            return self.ex(rhs)
        if rhs.kind == "Index" and n == 2:
            return f"go::map_get2({self.ex(rhs.x)}, {self.ex(rhs.index)})"
        if rhs.kind == "TypeAssert" and n == 2:
            return f"go::assert2<{self.ctype(rhs.typ)}>({self.ex(rhs.x)})"
        self.err(rhs, "expression does not yield several values")

    def unpack(self, names, rhs, define, all_new=False):
        t = self.tmpname()
        out = [f"auto {t} = {self.multi_value(rhs, len(names))};"]
        for i, nm in enumerate(names):
            if nm == "_":
                continue
            new = define and (all_new or nm not in self.scopes[-1])
            if new:
                out.append(f"auto& {mangle(nm)} = *go::New(std::get<{i}>({t}));" if self.heap(nm) else f"auto {mangle(nm)} = std::get<{i}>({t});")
            else:
                out.append(f"{mangle(nm)} = std::get<{i}>({t});")
        for nm in names:
            if nm != "_":
                self.declare(nm)
        return "\n".join(out)

    def st_Define(self, s):
        names, vals = s.names, s.values
        if len(vals) == 1 and len(names) > 1:
            return self.unpack(names, vals[0], define=True)
        if len(vals) != len(names):
            self.err(s, "define arity")
        exprs = []
        for v in vals:
            if v.kind == "CharLit":
                exprs.append(f"go::Rune({self.ex(v)})")
            else:
                exprs.append(self.ex(v))
        out = []
        if len(names) == 1:
            nm = names[0]
            if nm == "_":
                return f"(void)({exprs[0]});"
            if nm in self.scopes[-1]:
                return f"{mangle(nm)} = {exprs[0]};"
            used = set()
            walk_idents(vals[0], used)
            self.declare(nm)
            if nm in used:                                   # x := f(x) in a nested scope: the right side still means the OUTER x
                t = self.tmpname()
                pre = f"auto {t} = go::def({exprs[0]});\n"
                return pre + (f"auto& {mangle(nm)} = *go::New({t});" if self.heap(nm) else f"auto {mangle(nm)} = {t};")
            if self.heap(nm):
                return f"auto& {mangle(nm)} = *go::New(go::def({exprs[0]}));"
            return f"auto {mangle(nm)} = go::def({exprs[0]});"
        tmps = []
        for e in exprs:
            t = self.tmpname()
            tmps.append(t)
            out.append(f"auto {t} = go::def({e});")
        for nm, t in zip(names, tmps):
            if nm == "_":
                continue
            if nm in self.scopes[-1]:
                out.append(f"{mangle(nm)} = {t};")
            elif self.heap(nm):
                out.append(f"auto& {mangle(nm)} = *go::New({t});")
            else:
                out.append(f"auto {mangle(nm)} = {t};")
        for nm in names:
            if nm != "_":
                self.declare(nm)
        return "\n".join(out)

    def heap(self, name):
        return name in getattr(self, "escaping", ())

    def lhs(self, l):
        if l.kind == "Ident" and l.name == "_":
            return None
        return self.ex(l)

    def st_Assign(self, s):
        op = s.op
        if op != "=":
            if len(s.lhs) != 1 or len(s.rhs) != 1:
                self.err(s, "compound assignment arity")
            l, r = self.ex(s.lhs[0]), self.ex(s.rhs[0])
            if has_call(self, s.lhs[0]) and has_call(self, s.rhs[0]):
                # Go: the index operands on the left first, then the right side; C++17: the right side of an assignment first
                tl, trr = self.tmpname("o"), self.tmpname("o")
                body = f"{tl} = andnot({tl}, {trr});" if op == "&^=" else f"{tl} {op} {trr};"
                return f"{{ auto&& {tl} = {l}; auto {trr} = {r}; {body} }}"
            if op == "&^=":
                return f"{l} = andnot({l}, {r});"
            return f"{l} {op} {r};"
        if len(s.lhs) == 1 and len(s.rhs) == 1:
            l = self.lhs(s.lhs[0])
            r = self.ex(s.rhs[0])
            if l is not None and has_call(self, s.lhs[0]) and has_call(self, s.rhs[0]):
                tl, trr = self.tmpname("o"), self.tmpname("o")
                return f"{{ auto&& {tl} = {l}; auto {trr} = {r}; {tl} = {trr}; }}"
            return f"(void)({r});" if l is None else f"{l} = {r};"
        if len(s.rhs) == 1:
            t = self.tmpname()
            out = [f"auto {t} = {self.multi_value(s.rhs[0], len(s.lhs))};"]
            for i, l in enumerate(s.lhs):
                ls = self.lhs(l)
                if ls is not None:
                    out.append(f"{ls} = std::get<{i}>({t});")
            return "\n".join(out)
        if len(s.lhs) != len(s.rhs):
            self.err(s, "assignment arity")
        out, tmps = [], []
        for r in s.rhs:
            t = self.tmpname()
            tmps.append(t)
            out.append(f"auto {t} = {self.ex(r)};")
        for l, t in zip(s.lhs, tmps):
            ls = self.lhs(l)
            if ls is not None:
                out.append(f"{ls} = {t};")
        return "{ " + " ".join(out) + " }"

    def st_If(self, s):
        self.push()
        out = ["{"]
        if s.init is not None:
            out.append(self.stmt(s.init))
        out.append(f"if ({self.ex(s.cond)}) {self.block(s.body)}")
        if s.els is not None:
            out.append("else " + (self.st_If(s.els) if s.els.kind == "If" else self.block(s.els)))
        out.append("}")
        self.pop()
        return "\n".join(out)

    def loop_wrap(self, label, body_text_fn):
        """emit a loop body with the continue label of an enclosing `Label:` when a `continue Label` inside uses it"""
        saved = self.used_continue_labels
        self.used_continue_labels = set()
        text = body_text_fn()
        tail = ""
        if label is not None and label in self.used_continue_labels:
            tail = f"{mangle(label)}_continue:;"
        self.used_continue_labels = saved | (self.used_continue_labels - {label})
        return text, tail

    def st_For(self, s):
        label, self.loop_label = self.loop_label, None
        self.push()
        out = ["{"]
        if s.init is not None:
            if s.init.kind == "Define":
                bad = []
                closures_naming(s.body, set(s.init.names), bad)
                if bad:
                    self.err(bad[0], "a closure captures a loop variable (per-iteration copies of Go >= 1.22 are not translated)")
            out.append(self.stmt(s.init))
        cond = self.ex(s.cond) if s.cond is not None else ""
        post = ""
        if s.post is not None:
            p = self.stmt(s.post).rstrip()
            if "\n" in p or p.startswith("{") or p.count(";") > 1:
                post = "[&]{ " + p.replace("\n", " ") + " }()"
            else:
                post = p.rstrip(";")
        saved_fall = self.fall_var
        self.fall_var = None
        body, tail = self.loop_wrap(label, lambda: self.block(s.body))
        self.fall_var = saved_fall
        if tail:
            body = body[:body.rindex("}")] + tail + "\n}"
        out.append(f"for (; {cond}; {post}) {body}")
        out.append("}")
        self.pop()
        return "\n".join(out)

    def st_RangeFor(self, s):
        label, self.loop_label = self.loop_label, None
        captured = set()
        if s.define:
            # Go >= 1.22: every iteration has its own copy of the range variables. They are declared inside the loop body here, so that already holds;
            # one that a closure names gets heap (arena) storage per iteration, so that a closure that outlives the iteration keeps ITS copy.
            for nm in {x.name for x in (s.key, s.value) if x is not None and x.kind == "Ident"}:
                bad = []
                closures_naming(s.body, {nm}, bad)
                if bad:
                    captured.add(nm)
        self.push()
        r, n, i = self.tmpname("r"), self.tmpname("n"), self.tmpname("i")
        key = s.key if (s.key is not None and not (s.key.kind == "Ident" and s.key.name == "_")) else None
        val = s.value if (s.value is not None and not (s.value.kind == "Ident" and s.value.name == "_")) else None
        out = ["{", f"auto&& {r}x = {self.ex(s.x)};", f"auto {r} = go::{'ranger' if val is not None else 'ranger_keys'}({r}x);", f"int64_t {n} = {r}.n;"]
        head = []
        self.push()
        if key is not None:
            if s.define:
                self.declare(key.name)
                head.append(f"auto& {mangle(key.name)} = *go::New({r}.key({i}));" if key.name in captured else f"auto {mangle(key.name)} = {r}.key({i});")
            else:
                head.append(f"{self.ex(key)} = {r}.key({i});")
        if val is not None:
            if s.define:
                self.declare(val.name)
                head.append(f"auto& {mangle(val.name)} = *go::New({r}.val({i}));" if val.name in captured else f"auto {mangle(val.name)} = {r}.val({i});")
            else:
                head.append(f"{self.ex(val)} = {r}.val({i});")
        saved_fall = self.fall_var
        self.fall_var = None
        body, tail = self.loop_wrap(label, lambda: self.block(s.body))
        self.fall_var = saved_fall
        self.pop()
        out.append(f"for (int64_t {i} = 0; {i} < {n}; {i}++) {{")
        out.extend(head)
        out.append(body)
        if tail:
            out.append(tail)
        out.append("}")
        out.append("}")
        self.pop()
        return "\n".join(out)

    def st_Switch(self, s):
        label, self.loop_label = self.loop_label, None   # (a labelled switch: `break Label` leaves it)
        self.push()
        out = ["{"]
        if s.init is not None:
            out.append(self.stmt(s.init))
        tag = None
        if s.tag is not None:
            tag = self.tmpname("tag")
            out.append(f"auto&& {tag} = {self.ex(s.tag)};")
        sel = self.tmpname("sel")
        end = self.tmpname("next")
        clauses = s.clauses
        has_fall = any(c.body and c.body[-1].kind == "Branch" and c.body[-1].tok == "fallthrough" for c in clauses)
        # which clause runs
        out.append(f"int {sel} = -1;")
        first = True
        default_idx = None
        for idx, c in enumerate(clauses):
            if c.exprs is None:
                default_idx = idx
                continue
            conds = []
            for e in c.exprs:
                conds.append(f"({tag} == {self.ex(e)})" if tag is not None else f"({self.ex(e)})")
            out.append(f"{'if' if first else 'else if'} ({' || '.join(conds)}) {sel} = {idx};")
            first = False
        if default_idx is not None:
            out.append(f"{'else ' if not first else ''}{sel} = {default_idx};")
        out.append("switch (0) { default:")
        saved_fall = self.fall_var
        for idx, c in enumerate(clauses):
            lab = f"{end}_{idx}"
            self.fall_var = (sel, idx + 1, f"{end}_{idx + 1}") if has_fall else None
            self.push()
            body = "\n".join(self.stmt(b) for b in c.body)
            self.pop()
            pre = f"{lab}:;\n" if has_fall else ""
            out.append(f"{pre}if ({sel} == {idx}) {{\n{body}\n}}")
        if has_fall:
            out.append(f"{end}_{len(clauses)}:;")
        self.fall_var = saved_fall
        out.append("}")
        out.append("}")
        self.pop()
        return "\n".join(out)

    def st_TypeSwitch(self, s):
        label, self.loop_label = self.loop_label, None
        self.push()
        out = ["{"]
        if s.init is not None:
            out.append(self.stmt(s.init))
        ts = self.tmpname("ts")
        out.append(f"go::any {ts} = {self.ex(s.x)};")
        out.append("switch (0) { default:")
        first = True
        default = None
        saved_fall = self.fall_var
        self.fall_var = None
        for c in s.clauses:
            if c.types is None:
                default = c
                continue
            conds = []
            for t in c.types:
                if t.kind == "NamedType" and t.pkg is None and t.name == "nil":
                    conds.append(f"({ts} == go::nil)")
                else:
                    conds.append(f"go::type_is<{self.ctype(t)}>({ts})")
            self.push()
            bind = ""
            if s.bind and s.bind != "_":
                self.declare(s.bind)
                if len(c.types) == 1:
                    bind = f"auto {mangle(s.bind)} = go::assert1<{self.ctype(c.types[0])}>({ts}); (void){mangle(s.bind)};\n"
                else:
                    bind = f"auto& {mangle(s.bind)} = {ts}; (void){mangle(s.bind)};\n"
            body = "\n".join(self.stmt(b) for b in c.body)
            self.pop()
            out.append(f"{'if' if first else 'else if'} ({' || '.join(conds)}) {{\n{bind}{body}\n}}")
            first = False
        if default is not None:
            self.push()
            bind = ""
            if s.bind and s.bind != "_":
                self.declare(s.bind)
                bind = f"auto& {mangle(s.bind)} = {ts}; (void){mangle(s.bind)};\n"
            body = "\n".join(self.stmt(b) for b in default.body)
            self.pop()
            out.append(f"{'else ' if not first else ''}{{\n{bind}{body}\n}}")
        self.fall_var = saved_fall
        out.append("}")
        out.append("}")
        self.pop()
        return "\n".join(out)

    # ------------------------------------------------------------------------------------------------ functions
    def func_body(self, body, sig):
        if self.defer_frame and self.in_closure == 0:
            return self.func_body_deferring(body, sig)
        return self.func_body_plain(body, sig)

    def func_body_deferring(self, body, sig):
        """Go's defer / recover: the body runs inside try; deferred closures run after it, last in first out, whether it returned or panicked;
        recover() inside one of them stops the panic; results live outside the try block so that the closures can still change them."""
        self.push()
        rs = sig.results
        lines = ["{"]
        if rs and rs[0].name:
            names = []
            for r in rs:
                self.declare(r.name)
                lines.append(f"{self.ctype(r.typ)} {mangle(r.name)}{{}};")
                names.append(mangle(r.name))
        else:
            names = [f"_ret{i}" for i in range(len(rs))]
            for i, r in enumerate(rs):
                lines.append(f"{self.ctype(r.typ)} _ret{i}{{}};")
        lines.append("go::DeferFrame _df;")
        lines.append("try {")
        saved = (self.loop_label, self.fall_var, self.used_continue_labels)
        self.loop_label, self.fall_var, self.used_continue_labels = None, None, set()
        self.push()
        for st in body.stmts:
            lines.append(self.stmt(st))
        self.pop()
        self.loop_label, self.fall_var, self.used_continue_labels = saved
        lines.append("goto _df_done;")
        lines.append("} catch (go::PanicException& _e) { _df.set_panic(_e); }")
        lines.append("_df_done:;")
        lines.append("_df.run();")
        if not rs:
            lines.append("return;")
        elif len(rs) == 1:
            lines.append(f"return {names[0]};")
        else:
            lines.append("return std::tuple<" + ", ".join(self.ctype(r.typ) for r in rs) + ">(" + ", ".join(names) + ");")
        lines.append("}")
        self.pop()
        return "\n".join(lines)

    def func_body_plain(self, body, sig):
        """body block with the named results declared up front"""
        self.push()
        lines = ["{"]
        if sig.results and sig.results[0].name:
            for r in sig.results:
                if r.name != "_":
                    self.declare(r.name)
                    lines.append(f"{self.ctype(r.typ)} {mangle(r.name)}{{}};")
        saved = (self.loop_label, self.fall_var, self.used_continue_labels)
        self.loop_label, self.fall_var, self.used_continue_labels = None, None, set()
        for st in body.stmts:
            lines.append(self.stmt(st))
        self.loop_label, self.fall_var, self.used_continue_labels = saved
        if sig.results and not (body.stmts and body.stmts[-1].kind == "Return"):
            lines.append('go::panic_str("missing return");   // (Go proves the end unreachable, e.g. behind a switch whose every clause returns)')
        lines.append("}")
        self.pop()
        return "\n".join(lines)

    def begin_func(self, d):
        self.in_func = True
        self.scopes = [set()]
        self.ltypes = [{}]
        self.ltypes = [{}]
        self.cur_results = d.sig.results
        self.named_results = [r.name for r in d.sig.results] if d.sig.results and d.sig.results[0].name else None
        self.iota = None
        self.loop_label = None
        self.fall_var = None
        self.used_continue_labels = set()
        self.in_closure = 0
        self.defer_frame = has_defer(d.body)
        for p in d.sig.params:
            if p.name:
                self.declare(p.name)
        self.escaping = set()
        escaping_names(d.body, self.escaping)

    def func_def(self, d, owner=None):
        by_value = d.recv is not None and d.recv.typ.kind != "PointerType"       # a value receiver works on a COPY of the object
        THIS_IS_RECEIVER[0] = d.recv is not None and d.recv.name == "this" and not by_value
        self.begin_func(d)
        pre = ""
        if d.recv is not None and d.recv.name and d.recv.name != "_" and (by_value or d.recv.name != "this"):
            self.declare(d.recv.name)
            nm = mangle(d.recv.name)
            pre = f"auto {nm}_copy = *this;\nauto* {nm} = &{nm}_copy;\n" if by_value else f"auto* {nm} = this;\n"
        elif d.recv is not None and d.recv.name == "this":
            self.declare("this")
        heap_params = [p.name for p in d.sig.params if p.name and p.name != "_" and p.name in self.escaping and p.typ.kind not in ("SliceType", "PointerType")]
        self.heap_params = set(heap_params)
        body = self.func_body(d.body, d.sig)
        for hp in heap_params:
            pre += f"auto& {mangle(hp)} = *go::New({mangle(hp)}_arg);\n"
        if pre:
            body = "{\n" + pre + body[2:]
        name = mangle(d.name) if owner is None else f"{mangle(owner)}::{mangle(d.name)}"
        hoisted, self.hoisted = "".join(h + "\n" for h in self.hoisted), []
        return f"{hoisted}{self.result_ctype(d.sig)} {name}({self.params_decl(d.sig)}) {body}\n"

    def func_proto(self, d, in_class=False):
        self.scopes = [set()]
        self.ltypes = [{}]
        self.iota = None
        self.heap_params = set()
        return f"{self.result_ctype(d.sig)} {mangle(d.name)}({self.params_decl(d.sig)});"

    # ------------------------------------------------------------------------------------------------ package emission
    def method_sig(self, sig):
        return (tuple(self.param_ctype(p) for p in sig.params), tuple(self.ctype(r.typ) for r in sig.results))

    def file_of(self, pk, decl):
        for f in pk.files:
            if decl in f.decls:
                return f
        raise EmitError("declaration without a file")

    def emit_package(self, pk):
        out = [f"// ===== package {pk.name} ({pk.path}) =====", f"namespace {pk.ns} {{"]
        # `type A B` with B a struct of this package: A is a new struct type with B's fields (and its own methods)
        for t in pk.types.values():
            u = t.typ
            if u.kind == "NamedType" and u.pkg is None and not t.alias and u.name in pk.types and pk.types[u.name].typ.kind == "StructType":
                t.f["typ"] = pk.types[u.name].typ
        self.scopes = [set()]
        self.ltypes = [{}]
        self.iota = None
        self.cur_results, self.named_results = [], None
        self.loop_label, self.fall_var, self.used_continue_labels = None, None, set()
        structs = [t for t in pk.types.values() if t.typ.kind == "StructType"]
        ifaces = [t for t in pk.types.values() if t.typ.kind == "InterfaceType"]
        named = [t for t in pk.types.values() if t.typ.kind not in ("StructType", "InterfaceType")]
        for t in structs + ifaces:
            out.append(f"struct {mangle(t.name)};")
        for t in named:
            self.set_file(pk, self.file_of(pk, t))
            if t.typ.kind == "NamedType" and t.typ.pkg is None and t.typ.name in INT_RAW and not t.alias:
                out.append(f"struct {mangle(t.name)}_tag {{}};")
                out.append(f"using {mangle(t.name)} = go::I<{INT_RAW[t.typ.name]}, {mangle(t.name)}_tag>;")
            else:
                out.append(f"using {mangle(t.name)} = {self.ctype(t.typ)};")
        # constants and variables in dependency order
        items = {}
        for nm, (spec, i) in pk.consts.items():
            items[nm] = ("const", spec, i)
        for nm, (spec, i) in pk.vars.items():
            items[nm] = ("var", spec, i)
        deps = {}
        for nm, (kind, spec, i) in items.items():
            ids = set()
            if spec.values is not None:
                walk_idents(spec.values[i] if len(spec.values) == len(spec.names) else spec.values, ids)
            if spec.typ is not None:
                walk_idents(spec.typ, ids)
            # (Go orders package-level initialisation by references, also through the bodies of the functions an initialiser calls)
            seen_f, work = set(), [x for x in ids if x in pk.funcs]
            while work:
                fn = work.pop()
                if fn in seen_f:
                    continue
                seen_f.add(fn)
                sub = set()
                walk_idents(pk.funcs[fn].body, sub)
                ids |= sub
                work.extend(x for x in sub if x in pk.funcs and x not in seen_f)
            deps[nm] = {x for x in ids if x in items and x != nm}
        ordered, state = [], {}

        def visit(nm):
            if state.get(nm) == 2:
                return
            if state.get(nm) == 1:
                raise EmitError(f"initialisation cycle through {nm}")
            state[nm] = 1
            for d in sorted(deps[nm]):
                visit(d)
            state[nm] = 2
            ordered.append(nm)
        for nm in items:
            visit(nm)
        consts_out, vars_out = [], []
        self.in_func = False
        for nm in ordered:
            kind, spec, i = items[nm]
            if nm == "_":
                continue
            self.set_file(pk, self.file_of(pk, spec))
            if kind == "const":
                self.iota = spec.iota
                val = self.ex(spec.values[i])
                self.iota = None
                if spec.typ is not None:
                    consts_out.append(f"constexpr {self.ctype(spec.typ)} {mangle(nm)} = {val};")
                else:
                    v0 = spec.values[i]
                    if v0.kind == "StringLit":
                        consts_out.append(f"static const go::String {mangle(nm)} = {val};")
                    else:
                        consts_out.append(f"constexpr auto {mangle(nm)} = {val};")
            else:
                if spec.values is None:
                    vars_out.append(f"static {self.ctype(spec.typ)} {mangle(nm)}{{}};")
                elif len(spec.values) == len(spec.names):
                    v = self.value_for(spec.values[i], spec.typ)
                    if spec.typ is not None:
                        vars_out.append(f"static {self.ctype(spec.typ)} {mangle(nm)} = {v};")
                    else:
                        vars_out.append(f"static auto {mangle(nm)} = go::def({v});")
                else:
                    self.err(spec, "package-level multi-value var")
        out.extend(consts_out)
        # interfaces
        for t in ifaces:
            self.set_file(pk, self.file_of(pk, t))
            bases = [self.ctype(e).rstrip("*") for e in t.typ.embeds]
            lines = [f"struct {mangle(t.name)}" + (" : " + ", ".join("virtual " + b for b in bases) if bases else "") + " {"]
            sigs = {}
            for b in bases:
                sigs.update(self.iface_sigs.get(b, {}))
            for m in t.typ.methods:
                lines.append(f"  virtual {self.result_ctype(m.sig)} {mangle(m.name)}({self.params_decl(m.sig, names=False)}) = 0;")
                sigs[m.name] = self.method_sig(m.sig)
            lines.append(f"  virtual ~{mangle(t.name)}() = default;")
            lines.append("};")
            out.extend(lines)
            self.iface_sigs[f"::{pk.ns}::{mangle(t.name)}"] = sigs
        # structs, by-value containment first
        sdeps = {}
        sname = {t.name for t in structs}

        def value_deps(ty, acc):
            if ty.kind == "NamedType" and ty.pkg is None and ty.name in sname:
                acc.add(ty.name)
            elif ty.kind == "ArrayType":
                value_deps(ty.elem, acc)
        for t in structs:
            acc = set()
            for f in t.typ.fields:
                value_deps(f.typ, acc)
            sdeps[t.name] = acc
        sorder, sstate = [], {}

        def svisit(nm):
            if sstate.get(nm) == 2:
                return
            if sstate.get(nm) == 1:
                raise EmitError(f"struct {nm} contains itself")
            sstate[nm] = 1
            for d in sorted(sdeps[nm]):
                svisit(d)
            sstate[nm] = 2
            sorder.append(nm)
        for t in structs:
            svisit(t.name)
        for nm in sorder:
            t = pk.types[nm]
            self.set_file(pk, self.file_of(pk, t))
            methods = pk.methods.get(nm, [])
            msigs = {}
            for m in methods:
                self.set_file(pk, self.file_of(pk, m))
                msigs[m.name] = self.method_sig(m.sig)
            bases = []
            for iname, isig in self.iface_sigs.items():
                if isig and all(msigs.get(mn) == sg for mn, sg in isig.items()):
                    bases.append(iname)
            self.set_file(pk, self.file_of(pk, t))
            cn = mangle(nm)
            lines = [f"struct {cn}" + (" : " + ", ".join("virtual " + b for b in bases) if bases else "") + " {"]
            lines.append(f"  {cn}* operator->() {{ return this; }}")
            lines.append(f"  const {cn}* operator->() const {{ return this; }}")
            for f in t.typ.fields:
                if f.name == "_":
                    continue
                lines.append(f"  {self.ctype(f.typ)} {mangle(f.name)}{{}};")
            for m in methods:
                self.set_file(pk, self.file_of(pk, m))
                lines.append("  " + self.func_proto(m))
            lines.append("};")
            out.extend(lines)
        # prototypes
        for nm, d in pk.funcs.items():
            self.set_file(pk, self.file_of(pk, d))
            out.append(self.func_proto(d))
        for i, d in enumerate(pk.inits):
            out.append(f"void init_{i}();")
        out.extend(vars_out)
        # bodies
        for f in pk.files:
            self.set_file(pk, f)
            out.append(f"// ----- {f.pos[0]}")
            for d in f.decls:
                if d.kind != "FuncDecl" or d.body is None:
                    continue
                if d.recv is not None:
                    rt = d.recv.typ
                    tn = rt.elem.name if rt.kind == "PointerType" else rt.name
                    out.append(self.func_def(d, owner=tn))
                elif d.name == "init":
                    idx = pk.inits.index(d)
                    d2 = Node("FuncDecl", d.pos, name=f"init_{idx}", recv=None, sig=d.sig, body=d.body)
                    out.append(self.func_def(d2))
                else:
                    out.append(self.func_def(d))
        for i, d in enumerate(pk.inits):
            out.append(f"static const int init_done_{i} = (init_{i}(), 0);")
        # the package's own tests (files named *_test.go): registered by name for go_testing::run
        for f in pk.files:
            if not str(f.pos[0]).endswith("_test.go"):
                continue
            for d in f.decls:
                if d.kind == "FuncDecl" and d.recv is None and d.body is not None and d.name.startswith("Test") and len(d.sig.params) == 1:
                    out.append(f'static const ::go_testing::Register reg_{mangle(d.name)}("{pk.name}", "{d.name}", &{mangle(d.name)});')
        out.append(f"}}  // namespace {pk.ns}")
        return "\n".join(out)

    def emit_all(self):
        parts = ["// GENERATED by tools/go2cpp from the kanzi-go sources under /root/reference -- do not edit, do not commit (oracle/_ref/ is git-ignored).",
                 '#include "go_rt.hpp"', "using go::operator\"\"_u;", "using go::andnot;", ""]
        for path in self.order:
            parts.append(self.emit_package(self.pkgs[path]))
        return "\n".join(parts) + "\n"
