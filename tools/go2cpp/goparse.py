"""Lexer + recursive-descent parser for the subset of Go that kanzi-go's hot-path sources use.

TEST INFRASTRUCTURE (oracle/_ref): part of the mechanical Go -> C++ translation that builds a second, independent
CPU checker from the reference's own sources (see tools/go2cpp/README.md). Unknown syntax is a hard error: the
translator must never guess.

The AST is made of plain `Node` objects: Node(kind, **fields), `pos` = (file, line).
"""
import re


class GoSyntaxError(Exception):
    pass


class Node:
    __slots__ = ("kind", "pos", "f")

    def __init__(self, kind, pos=None, **f):
        self.kind = kind
        self.pos = pos
        self.f = f

    def __getattr__(self, k):
        try:
            return self.f[k]
        except KeyError:
            raise AttributeError(k)

    def __repr__(self):
        return f"Node({self.kind}, {self.f})"


KEYWORDS = {"break", "case", "chan", "const", "continue", "default", "defer", "else", "fallthrough", "for", "func", "go", "goto", "if",
            "import", "interface", "map", "package", "range", "return", "select", "struct", "switch", "type", "var"}

OPS = ["<<=", ">>=", "&^=", "...", "&&", "||", "<-", "++", "--", "==", "!=", "<=", ">=", ":=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=",
       "<<", ">>", "&^", "+", "-", "*", "/", "%", "&", "|", "^", "<", ">", "=", "!", "(", ")", "[", "]", "{", "}", ",", ";", ".", ":", "~"]

_ident_re = re.compile(r"[A-Za-z_-￿][A-Za-z0-9_-￿]*")
_num_re = re.compile(r"0[xX][0-9a-fA-F_]+|0[bB][01_]+|0[oO][0-7_]+|(?:[0-9][0-9_]*)?\.[0-9][0-9_]*(?:[eE][+-]?[0-9]+)?|[0-9][0-9_]*\.?(?:[eE][+-]?[0-9]+)?")


class Tok:
    __slots__ = ("kind", "val", "line")

    def __init__(self, kind, val, line):
        self.kind, self.val, self.line = kind, val, line

    def __repr__(self):
        return f"{self.kind}:{self.val!r}@{self.line}"


def _unescape(body, quote):
    """Go interpreted string / rune literal body -> bytes."""
    out = bytearray()
    i = 0
    while i < len(body):
        c = body[i]
        if c != "\\":
            out += c.encode("utf-8")
            i += 1
            continue
        i += 1
        e = body[i]
        simple = {"a": 7, "b": 8, "f": 12, "n": 10, "r": 13, "t": 9, "v": 11, "\\": 92, "'": 39, '"': 34}
        if e in simple:
            out.append(simple[e]); i += 1
        elif e == "x":
            out.append(int(body[i + 1:i + 3], 16)); i += 3
        elif e == "u":
            out += chr(int(body[i + 1:i + 5], 16)).encode("utf-8"); i += 5
        elif e == "U":
            out += chr(int(body[i + 1:i + 9], 16)).encode("utf-8"); i += 9
        elif e in "01234567":
            out.append(int(body[i:i + 3], 8)); i += 3
        else:
            raise GoSyntaxError(f"unknown escape \\{e}")
    return bytes(out)


def lex(src, fname):
    toks = []
    i, n, line = 0, len(src), 1
    last_sig = None     # last significant token on the current line (for semicolon insertion)

    def need_semi(t):
        if t is None:
            return False
        if t.kind in ("IDENT", "INT", "FLOAT", "CHAR", "STRING"):
            return True
        if t.kind == "KW" and t.val in ("break", "continue", "fallthrough", "return"):
            return True
        return t.kind == "OP" and t.val in ("++", "--", ")", "]", "}")

    while i < n:
        c = src[i]
        if c == "\n":
            if need_semi(last_sig):
                toks.append(Tok("OP", ";", line))
            last_sig = None
            line += 1; i += 1
            continue
        if c in " \t\r":
            i += 1
            continue
        if src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
            continue
        if src.startswith("/*", i):
            j = src.find("*/", i + 2)
            if j < 0:
                raise GoSyntaxError(f"{fname}:{line}: unterminated comment")
            seg = src[i:j + 2]
            if "\n" in seg and need_semi(last_sig):
                toks.append(Tok("OP", ";", line)); last_sig = None
            line += seg.count("\n")
            i = j + 2
            continue
        if c == "`":
            j = src.find("`", i + 1)
            body = src[i + 1:j].replace("\r", "")
            t = Tok("STRING", body.encode("utf-8"), line)
            line += body.count("\n")
            toks.append(t); last_sig = t; i = j + 1
            continue
        if c == '"':
            j = i + 1
            while src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            t = Tok("STRING", _unescape(src[i + 1:j], '"'), line)
            toks.append(t); last_sig = t; i = j + 1
            continue
        if c == "'":
            j = i + 1
            while src[j] != "'":
                j += 2 if src[j] == "\\" else 1
            body = src[i + 1:j]
            if body.startswith("\\") and body[1] in "x01234567":
                val = _unescape(body, "'")[0]
            else:
                val = ord(_unescape(body, "'").decode("utf-8"))
            t = Tok("CHAR", val, line)
            toks.append(t); last_sig = t; i = j + 1
            continue
        m = _ident_re.match(src, i)
        if m:
            w = m.group(0)
            t = Tok("KW" if w in KEYWORDS else "IDENT", w, line)
            toks.append(t); last_sig = t; i = m.end()
            continue
        if c.isdigit() or (c == "." and i + 1 < n and src[i + 1].isdigit()):
            m = _num_re.match(src, i)
            w = m.group(0).replace("_", "")
            lw = w.lower()
            if lw.startswith("0x"):
                t = Tok("INT", int(w, 16), line)
            elif lw.startswith("0b"):
                t = Tok("INT", int(w[2:], 2), line)
            elif lw.startswith("0o"):
                t = Tok("INT", int(w[2:], 8), line)
            elif "." in w or "e" in lw:
                t = Tok("FLOAT", float(w), line)
            elif len(w) > 1 and w[0] == "0":
                t = Tok("INT", int(w, 8), line)
            else:
                t = Tok("INT", int(w), line)
            toks.append(t); last_sig = t; i = m.end()
            continue
        for op in OPS:
            if src.startswith(op, i):
                t = Tok("OP", op, line)
                toks.append(t); last_sig = t; i += len(op)
                break
        else:
            raise GoSyntaxError(f"{fname}:{line}: unexpected character {c!r}")
    if need_semi(last_sig):
        toks.append(Tok("OP", ";", line))
    toks.append(Tok("EOF", None, line))
    return toks


BINPREC = {"||": 1, "&&": 2, "==": 3, "!=": 3, "<": 3, "<=": 3, ">": 3, ">=": 3, "+": 4, "-": 4, "|": 4, "^": 4,
           "*": 5, "/": 5, "%": 5, "<<": 5, ">>": 5, "&": 5, "&^": 5}
ASSIGN_OPS = {"=", ":=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "<<=", ">>=", "&^="}


class Parser:
    def __init__(self, src, fname):
        self.fname = fname
        self.toks = lex(src, fname)
        self.p = 0
        self.expr_lev = 0     # < 0: inside a control clause header (composite literals of bare type names not allowed)

    # ---- token helpers
    @property
    def t(self):
        return self.toks[self.p]

    def pos(self):
        return (self.fname, self.t.line)

    def err(self, msg):
        raise GoSyntaxError(f"{self.fname}:{self.t.line}: {msg} (at {self.t!r})")

    def is_op(self, *v):
        return self.t.kind == "OP" and self.t.val in v

    def is_kw(self, *v):
        return self.t.kind == "KW" and self.t.val in v

    def next(self):
        t = self.t
        self.p += 1
        return t

    def expect_op(self, v):
        if not self.is_op(v):
            self.err(f"expected {v!r}")
        return self.next()

    def expect_kw(self, v):
        if not self.is_kw(v):
            self.err(f"expected keyword {v!r}")
        return self.next()

    def accept_op(self, v):
        if self.is_op(v):
            self.next()
            return True
        return False

    def ident(self):
        if self.t.kind != "IDENT":
            self.err("expected identifier")
        return self.next().val

    def skip_semi(self):
        while self.is_op(";"):
            self.next()

    # ---- file
    def parse_file(self):
        self.skip_semi()
        self.expect_kw("package")
        pkg = self.ident()
        self.skip_semi()
        imports = []
        while self.is_kw("import"):
            self.next()
            if self.accept_op("("):
                while not self.is_op(")"):
                    imports.append(self.import_spec())
                    self.skip_semi()
                self.next()
            else:
                imports.append(self.import_spec())
            self.skip_semi()
        decls = []
        while self.t.kind != "EOF":
            decls.extend(self.top_decl())
            self.skip_semi()
        return Node("File", (self.fname, 1), package=pkg, imports=imports, decls=decls)

    def import_spec(self):
        alias = None
        if self.t.kind == "IDENT":
            alias = self.next().val
        elif self.is_op("."):
            self.err("dot imports are not supported")
        if self.t.kind != "STRING":
            self.err("expected import path")
        path = self.next().val.decode()
        return (alias or path.split("/")[-1], path)

    def top_decl(self):
        if self.is_kw("const", "var", "type"):
            return self.gen_decl()
        if self.is_kw("func"):
            return [self.func_decl()]
        self.err("unexpected top-level token")

    def gen_decl(self):
        kw = self.next().val
        specs = []
        parse = {"const": self.const_spec, "var": self.var_spec, "type": self.type_spec}[kw]
        if self.accept_op("("):
            idx = 0
            prev = None
            while not self.is_op(")"):
                s = parse(idx, prev)
                prev = s
                specs.append(s)
                idx += 1
                self.skip_semi()
            self.next()
        else:
            specs.append(parse(0, None))
        return specs

    def ident_list(self):
        names = [self.ident()]
        while self.accept_op(","):
            names.append(self.ident())
        return names

    def const_spec(self, iota, prev):
        pos = self.pos()
        names = self.ident_list()
        typ = None
        values = None
        if not self.is_op("=", ";", ")"):
            typ = self.parse_type()
        implicit = False
        if self.accept_op("="):
            values = self.expr_list()
        else:
            if prev is None:
                self.err("const without value")
            values, typ, implicit = prev.values, (typ or prev.typ), True
        return Node("ConstSpec", pos, names=names, typ=typ, values=values, iota=iota, implicit=implicit)

    def var_spec(self, idx=0, prev=None):
        pos = self.pos()
        names = self.ident_list()
        typ = None
        values = None
        if not self.is_op("="):
            typ = self.parse_type()
        if self.accept_op("="):
            values = self.expr_list()
        return Node("VarSpec", pos, names=names, typ=typ, values=values)

    def type_spec(self, idx=0, prev=None):
        pos = self.pos()
        name = self.ident()
        if self.is_op("["):
            # generic type parameters are not used by the translated files; "[N]T" array definitions are
            pass
        alias = self.accept_op("=")
        typ = self.parse_type()
        return Node("TypeSpec", pos, name=name, typ=typ, alias=alias)

    def func_decl(self):
        pos = self.pos()
        self.expect_kw("func")
        recv = None
        if self.is_op("("):
            params = self.param_list()
            if len(params) != 1:
                self.err("method receiver")
            recv = params[0]
        name = self.ident()
        if self.is_op("["):
            self.err("generic functions are not supported")
        sig = self.signature()
        body = None
        if self.is_op("{"):
            old = self.expr_lev
            self.expr_lev = 0
            body = self.block()
            self.expr_lev = old
        return Node("FuncDecl", pos, name=name, recv=recv, sig=sig, body=body)

    def signature(self):
        params = self.param_list()
        results = []
        if self.is_op("("):
            results = self.param_list()
        elif not self.is_op("{", ";", ")", ",", "]", "}", "=", ":=", ":") and self.t.kind != "EOF" and not (self.t.kind == "STRING"):
            results = [Node("Param", self.pos(), name=None, typ=self.parse_type(), variadic=False)]
        return Node("Signature", None, params=params, results=results)

    def param_list(self):
        self.expect_op("(")
        entries = []    # (maybe_name_or_type_expr, type_or_None, variadic)
        while not self.is_op(")"):
            pos = self.pos()
            variadic = False
            if self.accept_op("..."):
                variadic = True
                entries.append((pos, None, self.parse_type(), True))
            else:
                first = self.parse_type()
                if self.is_op(",", ")"):
                    entries.append((pos, None, first, False))
                else:
                    if first.kind != "NamedType" or first.pkg is not None:
                        self.err("parameter name expected")
                    if self.accept_op("..."):
                        variadic = True
                    entries.append((pos, first.name, self.parse_type(), variadic))
            if not self.accept_op(","):
                break
        self.expect_op(")")
        named = any(e[1] is not None for e in entries)
        out = []
        if not named:
            for pos, _, typ, var in entries:
                out.append(Node("Param", pos, name=None, typ=typ, variadic=var))
            return out
        # grouped: "a, b int" -> entries (None, a-as-type), (b, int): names without a type take the next type
        pending = []
        for pos, name, typ, var in entries:
            if name is None:
                if typ.kind != "NamedType" or typ.pkg is not None:
                    self.err("mixed named and unnamed parameters")
                pending.append((pos, typ.name))
            else:
                for ppos, pname in pending:
                    out.append(Node("Param", ppos, name=pname, typ=typ, variadic=False))
                pending = []
                out.append(Node("Param", pos, name=name, typ=typ, variadic=var))
        if pending:
            self.err("mixed named and unnamed parameters")
        return out

    # ---- types
    def parse_type(self):
        pos = self.pos()
        if self.t.kind == "IDENT":
            name = self.next().val
            if self.is_op(".") and self.toks[self.p + 1].kind == "IDENT":
                self.next()
                return Node("NamedType", pos, pkg=name, name=self.ident())
            return Node("NamedType", pos, pkg=None, name=name)
        if self.accept_op("*"):
            return Node("PointerType", pos, elem=self.parse_type())
        if self.accept_op("("):
            t = self.parse_type()
            self.expect_op(")")
            return t
        if self.accept_op("["):
            if self.accept_op("]"):
                return Node("SliceType", pos, elem=self.parse_type())
            if self.accept_op("..."):
                self.expect_op("]")
                return Node("ArrayType", pos, len=None, elem=self.parse_type())
            old = self.expr_lev
            self.expr_lev += 1
            ln = self.expr()
            self.expr_lev = old
            self.expect_op("]")
            return Node("ArrayType", pos, len=ln, elem=self.parse_type())
        if self.is_kw("map"):
            self.next()
            self.expect_op("[")
            k = self.parse_type()
            self.expect_op("]")
            return Node("MapType", pos, key=k, elem=self.parse_type())
        if self.is_kw("func"):
            self.next()
            return Node("FuncType", pos, sig=self.signature())
        if self.is_kw("struct"):
            return self.struct_type()
        if self.is_kw("interface"):
            return self.interface_type()
        self.err("type expected")

    def struct_type(self):
        pos = self.pos()
        self.expect_kw("struct")
        self.expect_op("{")
        fields = []
        while not self.is_op("}"):
            fpos = self.pos()
            if self.is_op("*") or (self.t.kind == "IDENT" and (self.toks[self.p + 1].kind == "OP" and self.toks[self.p + 1].val in (";", "}", "."))):
                self.err("embedded struct fields are not supported")
            names = self.ident_list()
            typ = self.parse_type()
            if self.t.kind == "STRING":
                self.next()    # struct tag
            for nm in names:
                fields.append(Node("Field", fpos, name=nm, typ=typ))
            self.skip_semi()
        self.next()
        return Node("StructType", pos, fields=fields)

    def interface_type(self):
        pos = self.pos()
        self.expect_kw("interface")
        self.expect_op("{")
        methods, embeds = [], []
        while not self.is_op("}"):
            mpos = self.pos()
            if self.t.kind == "IDENT" and self.toks[self.p + 1].kind == "OP" and self.toks[self.p + 1].val == "(":
                name = self.ident()
                methods.append(Node("Method", mpos, name=name, sig=self.signature()))
            else:
                embeds.append(self.parse_type())
            self.skip_semi()
        self.next()
        return Node("InterfaceType", pos, methods=methods, embeds=embeds)

    # ---- statements
    def block(self):
        pos = self.pos()
        self.expect_op("{")
        stmts = self.stmt_list()
        self.expect_op("}")
        return Node("Block", pos, stmts=stmts)

    def stmt_list(self):
        stmts = []
        self.skip_semi()
        while not self.is_op("}") and not self.is_kw("case", "default") and self.t.kind != "EOF":
            s = self.stmt()
            if s is not None:
                stmts.append(s)
            if self.is_op("}") or self.is_kw("case", "default"):
                break
            if not self.is_op(";"):
                self.err("expected ';' or newline after statement")
            self.skip_semi()
        return stmts

    def stmt(self):
        pos = self.pos()
        t = self.t
        if t.kind == "KW":
            v = t.val
            if v in ("var", "const", "type"):
                return Node("DeclStmt", pos, specs=self.gen_decl())
            if v == "return":
                self.next()
                vals = [] if self.is_op(";", "}") else self.expr_list()
                return Node("Return", pos, values=vals)
            if v in ("break", "continue", "goto"):
                self.next()
                label = self.next().val if self.t.kind == "IDENT" else None
                return Node("Branch", pos, tok=v, label=label)
            if v == "fallthrough":
                self.next()
                return Node("Branch", pos, tok=v, label=None)
            if v == "if":
                return self.if_stmt()
            if v == "for":
                return self.for_stmt()
            if v == "switch":
                return self.switch_stmt()
            if v == "go":
                self.next()
                return Node("Go", pos, call=self.expr())
            if v == "defer":
                self.next()
                return Node("Defer", pos, call=self.expr())
            if v == "func":
                return self.simple_stmt()
            self.err(f"statement keyword {v!r} is not supported")
        if self.is_op("{"):
            return self.block()
        if self.is_op(";"):
            return None
        if t.kind == "IDENT" and self.toks[self.p + 1].kind == "OP" and self.toks[self.p + 1].val == ":" :
            name = self.next().val
            self.next()
            self.skip_semi()
            inner = None if self.is_op("}") else self.stmt()
            return Node("Labeled", pos, label=name, stmt=inner)
        return self.simple_stmt()

    def simple_stmt(self, range_ok=False):
        pos = self.pos()
        if range_ok and self.is_kw("range"):
            self.next()
            return Node("Range", pos, key=None, value=None, define=False, x=self.expr())
        lhs = self.expr_list()
        if self.is_op("++", "--"):
            op = self.next().val
            return Node("IncDec", pos, x=lhs[0], op=op)
        if self.t.kind == "OP" and self.t.val in ASSIGN_OPS:
            op = self.next().val
            if range_ok and self.is_kw("range"):
                self.next()
                x = self.expr()
                return Node("Range", pos, key=lhs[0], value=lhs[1] if len(lhs) > 1 else None, define=(op == ":="), x=x)
            rhs = self.expr_list()
            if op == ":=":
                for l in lhs:
                    if l.kind != "Ident":
                        self.err("non-name on left side of :=")
                return Node("Define", pos, names=[l.name for l in lhs], values=rhs)
            return Node("Assign", pos, lhs=lhs, op=op, rhs=rhs)
        if len(lhs) != 1:
            self.err("expression list as statement")
        return Node("ExprStmt", pos, x=lhs[0])

    def if_stmt(self):
        pos = self.pos()
        self.expect_kw("if")
        old = self.expr_lev
        self.expr_lev = -1
        init = None
        cond = None
        if self.is_op(";"):
            self.next()
            cond = self.expr()
        else:
            s = self.simple_stmt()
            if self.accept_op(";"):
                init = s
                cond = self.expr()
            else:
                if s.kind != "ExprStmt":
                    self.err("if condition")
                cond = s.x
        self.expr_lev = old
        body = self.block()
        els = None
        if self.is_kw("else"):
            self.next()
            els = self.if_stmt() if self.is_kw("if") else self.block()
        return Node("If", pos, init=init, cond=cond, body=body, els=els)

    def for_stmt(self):
        pos = self.pos()
        self.expect_kw("for")
        old = self.expr_lev
        self.expr_lev = -1
        init = cond = post = None
        rng = None
        if not self.is_op("{"):
            if self.is_op(";"):
                pass
            else:
                s = self.simple_stmt(range_ok=True)
                if s.kind == "Range":
                    rng = s
                elif self.is_op("{"):
                    if s.kind != "ExprStmt":
                        self.err("for condition")
                    cond = s.x
                else:
                    init = s
            if rng is None and cond is None and not self.is_op("{"):
                self.expect_op(";")
                if not self.is_op(";"):
                    cond = self.expr()
                self.expect_op(";")
                if not self.is_op("{"):
                    post = self.simple_stmt()
        self.expr_lev = old
        body = self.block()
        if rng is not None:
            return Node("RangeFor", pos, key=rng.key, value=rng.value, define=rng.define, x=rng.x, body=body)
        return Node("For", pos, init=init, cond=cond, post=post, body=body)

    def switch_stmt(self):
        pos = self.pos()
        self.expect_kw("switch")
        old = self.expr_lev
        self.expr_lev = -1
        init = None
        tag = None
        if not self.is_op("{"):
            s = None
            if not self.is_op(";"):
                s = self.simple_stmt()
            if self.accept_op(";"):
                init = s
                if not self.is_op("{"):
                    s2 = self.simple_stmt()
                    if s2.kind != "ExprStmt":
                        self.err("switch tag")
                    tag = s2.x
            else:
                if s.kind == "Define" and len(s.names) == 1 and len(s.values) == 1 and s.values[0].kind == "TypeAssert" and s.values[0].typ is None:
                    return self.type_switch_body(pos, init, s.names[0], s.values[0].x, old)      # switch v := x.(type)
                if s.kind != "ExprStmt":
                    self.err("switch header")
                tag = s.x
                if tag.kind == "TypeAssert" and tag.typ is None:
                    return self.type_switch_body(pos, init, None, tag.x, old)
        self.expr_lev = old
        self.expect_op("{")
        clauses = []
        while not self.is_op("}"):
            cpos = self.pos()
            if self.is_kw("default"):
                self.next()
                exprs = None
            else:
                self.expect_kw("case")
                exprs = self.expr_list()
            self.expect_op(":")
            body = self.stmt_list()
            clauses.append(Node("Case", cpos, exprs=exprs, body=body))
        self.next()
        return Node("Switch", pos, init=init, tag=tag, clauses=clauses)

    def type_switch_body(self, pos, init, bind, x, old_lev):
        self.expr_lev = old_lev
        self.expect_op("{")
        clauses = []
        while not self.is_op("}"):
            cpos = self.pos()
            if self.is_kw("default"):
                self.next()
                types = None
            else:
                self.expect_kw("case")
                types = [self.parse_type()]
                while self.accept_op(","):
                    types.append(self.parse_type())
            self.expect_op(":")
            body = self.stmt_list()
            clauses.append(Node("TypeCase", cpos, types=types, body=body))
        self.next()
        return Node("TypeSwitch", pos, init=init, bind=bind, x=x, clauses=clauses)

    # ---- expressions
    def expr_list(self):
        xs = [self.expr()]
        while self.accept_op(","):
            xs.append(self.expr())
        return xs

    def expr(self, prec=1):
        x = self.unary()
        while self.t.kind == "OP" and self.t.val in BINPREC and BINPREC[self.t.val] >= prec:
            pos = self.pos()
            op = self.next().val
            y = self.expr(BINPREC[op] + 1)
            x = Node("Binary", pos, op=op, x=x, y=y)
        return x

    def unary(self):
        pos = self.pos()
        if self.t.kind == "OP" and self.t.val in ("+", "-", "!", "^", "*", "&"):
            op = self.next().val
            x = self.unary()
            return Node("Unary", pos, op=op, x=x)
        if self.is_op("<-"):
            self.err("channel operations are not supported")
        return self.primary()

    def is_type_start(self):
        return self.is_op("[") or self.is_kw("map", "struct", "func", "interface", "chan")

    def primary(self):
        pos = self.pos()
        t = self.t
        if t.kind == "INT":
            self.next(); x = Node("IntLit", pos, value=t.val)
        elif t.kind == "FLOAT":
            self.next(); x = Node("FloatLit", pos, value=t.val)
        elif t.kind == "CHAR":
            self.next(); x = Node("CharLit", pos, value=t.val)
        elif t.kind == "STRING":
            self.next(); x = Node("StringLit", pos, value=t.val)
        elif t.kind == "IDENT":
            self.next(); x = Node("Ident", pos, name=t.val)
        elif self.is_op("("):
            self.next()
            old = self.expr_lev
            self.expr_lev += 1 if self.expr_lev >= 0 else 1 - self.expr_lev
            if self.is_op("*") or self.is_type_start():
                # could be a parenthesised type for a conversion: (*T)(x) ; try expression first (unary * works for both)
                inner = self.expr_or_type()
            else:
                inner = self.expr()
            self.expr_lev = old
            self.expect_op(")")
            x = Node("Paren", pos, x=inner)
        elif self.is_kw("func"):
            self.next()
            sig = self.signature()
            if self.is_op("{"):
                old = self.expr_lev
                self.expr_lev = 0
                body = self.block()
                self.expr_lev = old
                x = Node("FuncLit", pos, sig=sig, body=body)
            else:
                x = Node("TypeExpr", pos, typ=Node("FuncType", pos, sig=sig))
        elif self.is_type_start():
            typ = self.parse_type()
            x = Node("TypeExpr", pos, typ=typ)
        else:
            self.err("operand expected")
        # suffixes
        while True:
            pos = self.pos()
            if self.is_op("."):
                self.next()
                if self.accept_op("("):
                    if self.is_kw("type"):
                        self.next()
                        typ = None
                    else:
                        typ = self.parse_type()
                    self.expect_op(")")
                    x = Node("TypeAssert", pos, x=x, typ=typ)
                else:
                    x = Node("Selector", pos, x=x, sel=self.ident())
            elif self.is_op("["):
                self.next()
                old = self.expr_lev
                self.expr_lev += 1 if self.expr_lev >= 0 else 1 - self.expr_lev
                idx = [None, None, None]
                ncolon = 0
                if not self.is_op(":"):
                    idx[0] = self.expr()
                while self.accept_op(":"):
                    ncolon += 1
                    if not self.is_op(":", "]"):
                        idx[ncolon] = self.expr()
                self.expr_lev = old
                self.expect_op("]")
                if ncolon == 0:
                    x = Node("Index", pos, x=x, index=idx[0])
                else:
                    x = Node("SliceExpr", pos, x=x, lo=idx[0], hi=idx[1], max=idx[2], three=(ncolon == 2))
            elif self.is_op("("):
                self.next()
                old = self.expr_lev
                self.expr_lev += 1 if self.expr_lev >= 0 else 1 - self.expr_lev
                args = []
                spread = False
                while not self.is_op(")"):
                    if self.is_type_start() and not self.is_kw("func"):
                        tpos = self.pos()
                        typ = self.parse_type()
                        a = Node("TypeExpr", tpos, typ=typ)
                        if self.is_op("{"):
                            a = self.composite_body(a, tpos)
                        elif self.is_op("("):
                            # conversion []byte(x) used as an argument
                            self.next()
                            inner = self.expr()
                            self.expect_op(")")
                            a = Node("Call", tpos, fun=a, args=[inner], spread=False)
                        a = self.suffix_continue(a)
                        args.append(a)
                    else:
                        args.append(self.expr())
                    if self.accept_op("..."):
                        spread = True
                    if not self.accept_op(","):
                        break
                self.expr_lev = old
                self.expect_op(")")
                x = Node("Call", pos, fun=x, args=args, spread=spread)
            elif self.is_op("{") and self.composite_ok(x):
                x = self.composite_body(x, pos)
            else:
                return x

    def suffix_continue(self, a):
        """continue binary expression parsing after a primary that started with a type (rare)."""
        while self.t.kind == "OP" and self.t.val in BINPREC:
            pos = self.pos()
            op = self.next().val
            y = self.expr(BINPREC[op] + 1)
            a = Node("Binary", pos, op=op, x=a, y=y)
        return a

    def expr_or_type(self):
        # "(*T)(x)": unary parses "*T" as a deref of identifier T; the emitter resolves T as a type
        return self.expr()

    def composite_ok(self, x):
        if x.kind == "TypeExpr":
            return True
        if self.expr_lev < 0:
            return False
        if x.kind == "Ident":
            return True
        if x.kind == "Selector" and x.x.kind == "Ident":
            return True
        return False

    def composite_body(self, typ_expr, pos):
        self.expect_op("{")
        old = self.expr_lev
        self.expr_lev = 0
        elems = []
        self.skip_semi()
        while not self.is_op("}"):
            elems.append(self.element())
            self.skip_semi()
            if not self.accept_op(","):
                self.skip_semi()
                break
            self.skip_semi()
        self.expr_lev = old
        self.expect_op("}")
        return Node("Composite", pos, typ=typ_expr, elems=elems)

    def element(self):
        pos = self.pos()
        v = self.element_value()
        if self.accept_op(":"):
            return Node("KeyValue", pos, key=v, value=self.element_value())
        return Node("KeyValue", pos, key=None, value=v)

    def element_value(self):
        if self.is_op("{"):
            pos = self.pos()
            return self.composite_body(None, pos)
        return self.expr()


def parse_file(path):
    with open(path, encoding="utf-8") as f:
        src = f.read()
    return Parser(src, path).parse_file()


if __name__ == "__main__":
    import sys
    for p in sys.argv[1:]:
        f = parse_file(p)
        print(p, f.package, len(f.decls), "declarations")
