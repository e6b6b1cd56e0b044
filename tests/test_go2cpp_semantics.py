"""The translator itself (tools/go2cpp), against the Go specification: tests/golden/go2cpp/semantics.go holds small Go functions whose results follow from
the language rules the kanzi-go sources lean on (shifts >= width, wrap-around, arithmetic >>, untyped constants, Go's operator precedence, truncating
division, slices as views with append / copy / 3-index semantics, arrays as values, parallel assignment, shadowing, named results, switch / fallthrough /
labelled break and continue / goto, defer order, recover of string and runtime panics through a type switch, structural interfaces, closures, maps as
references whose reads do not insert, strings as bytes, calls inside one expression run left to right, function-local and anonymous struct types, map literals, per-iteration range
variables under closures, type assertions on non-empty interfaces and on elements of maps of interfaces). It is translated and compiled here (g++), and every result must be the value Go gives."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G2C = os.path.join(ROOT, "tools", "go2cpp")

MAIN = r'''
#include "semantics.gen.hpp"
#include <cstdio>
using namespace kz_semantics;
static void p(long long v) { std::printf("%lld ", v); }
static void pu(unsigned long long v) { std::printf("%llu ", v); }
static void ps(const go::String& s) { std::printf("[%s] ", s.s.c_str()); }
int main() {
    { auto [a, b, c, d, e] = Shifts(go::Uint(go::U(64))); std::printf("shifts "); pu(a.v); pu(b.v); p(c.v); p(d.v); pu(e.v); std::printf("\n"); }
    { auto [y, m, b, z, w] = Consts(); std::printf("consts "); p(y.v); pu(m.v); pu(b.v); p(z.v); pu(w.v); std::printf("\n"); }
    { auto [pp, q, r, s] = Precedence(go::Uint32(go::U(0x12345678)), go::Uint32(go::U(0x0F0F00FF))); std::printf("prec "); pu(pp.v); pu(q.v); p(r ? 1 : 0); pu(s.v); std::printf("\n"); }
    { auto [a, b, c, d, e, f, g] = Arithmetic(); std::printf("arith "); p(a.v); p(b.v); p(c.v); pu(d.v); p(e.v); pu(f.v); p(g.v); std::printf("\n"); }
    { auto [a, b, c, d, e, f, g] = Slices(); std::printf("slices "); p(a.v); p(b.v); pu(c.v); pu(d.v); p(e.v); p(f.v); pu(g.v); std::printf("\n"); }
    { auto [a, b, c] = Arrays(); std::printf("arrays "); p(a.v); p(b.v); p(c.v); std::printf("\n"); }
    for (int n : {0, 3, 9}) { auto [x, y, err] = Scopes(go::Int(go::U(n))); std::printf("scopes "); p(x.v); p(y.v); ps(err ? err->Error() : go::String("nil")); std::printf("\n"); }
    { auto [a, b, c] = Control(go::Int(go::U(5))); std::printf("control "); p(a.v); p(b.v); p(c.v); std::printf("\n"); }
    for (int t : {0, 1, 2}) { auto [res, msg] = Deferred(go::Int(go::U(t))); std::printf("deferred "); p(res.v); ps(msg); std::printf("\n"); }
    { auto [a, b, c] = Interfaces(); std::printf("ifaces "); p(a.v); ps(b); p(c.v); std::printf("\n"); }
    { auto [a, b, c, d, e, f, g, h] = Values(go::Uint(go::U(40))); std::printf("values "); p(a.v); p(b.v); p(c.v); p(d.v); p(e.v); p(f.v); pu(g.v); p(h.v); std::printf("\n"); }
    { auto [a, ok, n, c, s, l] = MapsAndStrings(); std::printf("maps "); p(a.v); p(ok ? 1 : 0); p(n.v); pu(c.v); ps(s); p(l.v); std::printf("\n"); }
    { auto [a, okA, okB, okC] = MapAsserts(); std::printf("mapassert "); p(a.v); p(okA ? 1 : 0); p(okB ? 1 : 0); p(okC ? 1 : 0); std::printf("\n"); }
    { auto [a, l1, b, l2, l3, xy, l4] = CallOrder(); std::printf("order "); p(a.v); p(l1.v); p(b.v); p(l2.v); p(l3.v); p(xy.v); p(l4.v); std::printf("\n"); }
    { auto [a, b, c, d, e, f] = TableDriven(); std::printf("table "); p(a.v); p(b.v); p(c.v); p(d.v); p(e ? 1 : 0); ps(f); std::printf("\n"); }
    { std::printf("formats "); ps(Formats()); std::printf("\n"); }
    return 0;
}
'''

# what the Go specification gives for tests/golden/go2cpp/semantics.go (each line: the comments in that file say why)
EXPECTED = """shifts 0 2 -4 -1 44
consts 5497558138880 18446744073709551615 16 1125899906842624 4294967295
prec 305419911 1474321119 0 397923967
arith -3 -1 1 240 -56 18446744073709551615 7
slices 3 7 99 3 3 6 3
arrays 1 100 30
scopes 2 3 [nil]
scopes 2 21 [nil]
scopes 2 21 [big]
control 3 1 33
deferred 11 [fine]
deferred 101 [string:boom]
deferred 101 [error:runtime error:]
ifaces 37 [rs] 5
values 11 11 111 7 41 303 4294967295 1099511627776
maps 7 0 6 195 [51-x] 1
mapassert 6 1 0 0
order 7 123 463 4567 1234055 89 89
table 620 21 221 34 0 [xyb]
formats [[00011111] [abcdef] [FF] [ -42] [7 ] [-0042] ["hi"] [3] [50%]]
"""


def test_translated_go_semantics(tmp_path):
    gen = tmp_path / "semantics.gen.hpp"
    subprocess.check_call([sys.executable, os.path.join(G2C, "go2cpp.py"), "--module", "example.com/t", "--out", str(gen),
                           "semantics=" + os.path.join(ROOT, "tests", "golden", "go2cpp", "semantics.go")])
    (tmp_path / "main.cpp").write_text(MAIN)
    exe = tmp_path / "semantics"
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-fwrapv", "-I", os.path.join(G2C, "runtime"), "-I", str(tmp_path), "-o", str(exe), str(tmp_path / "main.cpp")])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, text=True, check=True).stdout
    got = [" ".join(l.split()) for l in out.strip().splitlines()]
    want = [" ".join(l.split()) for l in EXPECTED.strip().splitlines()]
    assert got == want, "\n".join(f"{g!r:60} | {w!r}" for g, w in zip(got, want) if g != w)


def test_translator_refuses_what_it_does_not_carry(tmp_path):
    """unknown or untranslated semantics are a hard error, never a guess: a closure over a loop variable (per-iteration copies since Go 1.22), a channel, a select"""
    cases = {
        "loopvar": "package p\nfunc F() []func() int {\n\tvar fs []func() int\n\tfor i := 0; i < 3; i++ {\n\t\tfs = append(fs, func() int { return i })\n\t}\n\treturn fs\n}\n",
        "channel": "package p\nfunc F(c chan int) int {\n\treturn <-c\n}\n",
        "generic": "package p\nfunc F[T any](x T) T {\n\treturn x\n}\n",
    }
    for name, src in cases.items():
        f = tmp_path / (name + ".go")
        f.write_text(src)
        r = subprocess.run([sys.executable, os.path.join(G2C, "go2cpp.py"), "--module", "example.com/t", "--out", str(tmp_path / "o.hpp"), "p=" + str(f)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        assert r.returncode != 0 and "go2cpp:" in r.stderr, (name, r.stdout, r.stderr)
