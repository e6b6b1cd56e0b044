// Go semantics for the C++ that tools/go2cpp emits from kanzi-go's sources (oracle/_ref).
//
// TEST INFRASTRUCTURE. Everything here is a RUNTIME SHIM, i.e. the part of "Go" the emitter does not generate: sized integers
// with Go's wrap-around / shift rules, untyped constants, slices / arrays with bounds checks that panic, strings, maps,
// `any`, `error`, panics as C++ exceptions, and the handful of standard-library calls the translated files make
// (encoding/binary, math/bits, errors, fmt, sort, slices, io, sync). No algorithm of the reference lives in this file.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <typeinfo>
#include <utility>
#include <vector>

namespace go {

// ---------------------------------------------------------------------------------------------------------------- panics
// a panic is a C++ exception. `is_error`: the panic value implements `error` (runtime errors do, as in Go: a recovered index-out-of-range
// lands in the `case error:` arm of the reference's recover blocks); otherwise it is a string.
struct PanicException : std::runtime_error {
    bool is_error;
    void* err_obj = nullptr;       // panic(err): the error value itself (recover() hands the same value back, so that errors.Is(err, io.EOF) holds)
    explicit PanicException(const std::string& m, bool is_err = true, void* obj = nullptr) : std::runtime_error(m), is_error(is_err), err_obj(obj) {}
};
[[noreturn]] inline void panic_str(const std::string& m) { throw PanicException(m, true); }

// ---------------------------------------------------------------------------------------------------------------- memory
// Allocations of the translated code live until the C API call that made them returns (Go has a garbage collector; a test
// checker has scopes). Outside any scope (static initialisers: package-level tables) memory is never freed.
struct Arena {
    std::vector<void*> blocks;
    ~Arena() { for (void* p : blocks) std::free(p); }
};
inline thread_local Arena* g_arena = nullptr;
struct ArenaScope {
    Arena a; Arena* prev;
    ArenaScope() : prev(g_arena) { g_arena = &a; }
    ~ArenaScope() { g_arena = prev; }
};
inline void* alloc_zero(size_t n) {
    void* p = std::calloc(n ? n : 1, 1);
    if (!p) panic_str("out of memory");
    if (g_arena) g_arena->blocks.push_back(p);
    return p;
}

// ---------------------------------------------------------------------------------------------------------------- nil
struct nil_t {
    template <class T> constexpr operator T*() const { return nullptr; }
};
inline constexpr nil_t nil{};
template <class T> inline bool operator==(T* p, nil_t) { return p == nullptr; }
template <class T> inline bool operator!=(T* p, nil_t) { return p != nullptr; }
template <class T> inline bool operator==(nil_t, T* p) { return p == nullptr; }
template <class T> inline bool operator!=(nil_t, T* p) { return p != nullptr; }

// ---------------------------------------------------------------------------------------------------------------- integers
// untyped integer constant: arbitrary precision in Go, 128 bits here (the translated files never exceed 65 bits)
struct U {
    __int128 v;
    constexpr U() : v(0) {}
    constexpr U(__int128 x) : v(x) {}
    constexpr explicit operator double() const { return (double)v; }
};
constexpr U operator""_u(unsigned long long x) { return U((__int128)x); }
#define GO_U_BIN(op) constexpr U operator op(U a, U b) { return U(a.v op b.v); }
GO_U_BIN(+) GO_U_BIN(-) GO_U_BIN(*) GO_U_BIN(&) GO_U_BIN(|) GO_U_BIN(^)
#undef GO_U_BIN
constexpr U operator/(U a, U b) { return U(a.v / b.v); }
constexpr U operator%(U a, U b) { return U(a.v % b.v); }
constexpr U operator<<(U a, U b) { return U(b.v >= 127 ? (__int128)0 : (__int128)((unsigned __int128)a.v << (int)b.v)); }
constexpr U operator>>(U a, U b) { return U(b.v >= 127 ? (a.v < 0 ? (__int128)-1 : (__int128)0) : a.v >> (int)b.v); }
constexpr U operator-(U a) { return U(-a.v); }
constexpr U operator+(U a) { return a; }
constexpr U operator~(U a) { return U(~a.v); }
#define GO_U_CMP(op) constexpr bool operator op(U a, U b) { return a.v op b.v; }
GO_U_CMP(==) GO_U_CMP(!=) GO_U_CMP(<) GO_U_CMP(<=) GO_U_CMP(>) GO_U_CMP(>=)
#undef GO_U_CMP
constexpr U andnot(U a, U b) { return U(a.v & ~b.v); }
// untyped constants and float64
constexpr double operator+(U a, double b) { return (double)a.v + b; }
constexpr double operator+(double a, U b) { return a + (double)b.v; }
constexpr double operator-(U a, double b) { return (double)a.v - b; }
constexpr double operator-(double a, U b) { return a - (double)b.v; }
constexpr double operator*(U a, double b) { return (double)a.v * b; }
constexpr double operator*(double a, U b) { return a * (double)b.v; }
constexpr double operator/(U a, double b) { return (double)a.v / b; }
constexpr double operator/(double a, U b) { return a / (double)b.v; }
#define GO_UD_CMP(op) constexpr bool operator op(U a, double b) { return (double)a.v op b; } constexpr bool operator op(double a, U b) { return a op (double)b.v; }
GO_UD_CMP(==) GO_UD_CMP(!=) GO_UD_CMP(<) GO_UD_CMP(<=) GO_UD_CMP(>) GO_UD_CMP(>=)
#undef GO_UD_CMP

template <class T, class Tag = void> struct I;
template <class X> struct is_goint : std::false_type {};
template <class T, class Tag> struct is_goint<I<T, Tag>> : std::true_type {};

// shift count of any integer type (Go: a negative count panics)
template <class S> constexpr uint64_t shift_count(S s) {
    if constexpr (std::is_same_v<S, U>) { if (s.v < 0) panic_str("negative shift amount"); return s.v > 1000 ? 1000 : (uint64_t)s.v; }
    else { if constexpr (std::is_signed_v<decltype(s.v)>) { if (s.v < 0) panic_str("negative shift amount"); } return (uint64_t)s.v; }
}

template <class T, class Tag> struct I {
    using raw = T;
    using UT = std::make_unsigned_t<T>;
    static constexpr int BITS = 8 * (int)sizeof(T);
    T v;
    constexpr I() : v(0) {}
    constexpr I(U u) : v((T)(UT)(unsigned __int128)u.v) {}                                          // untyped constant -> typed (implicit, as in Go)
    template <class T2, class Tag2> constexpr explicit I(I<T2, Tag2> o) : v((T)o.v) {}                 // T(x): truncation / sign extension as in Go
    constexpr explicit I(double d) : v((T)d) {}
    static constexpr I from_raw(T x) { I r; r.v = x; return r; }
    constexpr explicit operator double() const { return (double)v; }
    constexpr explicit operator float() const { return (float)v; }

    // arithmetic wraps (computed in 64-bit unsigned, truncated back)
    friend constexpr I operator+(I a, I b) { return from_raw((T)((uint64_t)a.v + (uint64_t)b.v)); }
    friend constexpr I operator-(I a, I b) { return from_raw((T)((uint64_t)a.v - (uint64_t)b.v)); }
    friend constexpr I operator*(I a, I b) { return from_raw((T)((uint64_t)a.v * (uint64_t)b.v)); }
    friend constexpr I operator/(I a, I b) {
        if (b.v == 0) panic_str("integer divide by zero");
        if constexpr (std::is_signed_v<T>) { if (b.v == (T)-1) return from_raw((T)(0 - (uint64_t)a.v)); }
        return from_raw((T)(a.v / b.v));
    }
    friend constexpr I operator%(I a, I b) {
        if (b.v == 0) panic_str("integer divide by zero");
        if constexpr (std::is_signed_v<T>) { if (b.v == (T)-1) return from_raw(0); }
        return from_raw((T)(a.v % b.v));
    }
    friend constexpr I operator&(I a, I b) { return from_raw((T)(a.v & b.v)); }
    friend constexpr I operator|(I a, I b) { return from_raw((T)(a.v | b.v)); }
    friend constexpr I operator^(I a, I b) { return from_raw((T)(a.v ^ b.v)); }
    friend constexpr I andnot(I a, I b) { return from_raw((T)(a.v & ~b.v)); }
    constexpr I operator-() const { return from_raw((T)(0 - (uint64_t)v)); }
    constexpr I operator+() const { return *this; }
    constexpr I operator~() const { return from_raw((T)~v); }
    // shifts: count >= width gives 0 (or the sign for an arithmetic right shift)
    template <class S> constexpr I shl(S s) const { uint64_t c = shift_count(s); return c >= (uint64_t)BITS ? from_raw(0) : from_raw((T)((UT)v << c)); }
    template <class S> constexpr I shr(S s) const {
        uint64_t c = shift_count(s);
        if (c >= (uint64_t)BITS) return from_raw(std::is_signed_v<T> && v < 0 ? (T)-1 : (T)0);
        return from_raw((T)(v >> c));
    }
    template <class T2, class G2> friend constexpr I operator<<(I a, I<T2, G2> s) { return a.shl(s); }
    template <class T2, class G2> friend constexpr I operator>>(I a, I<T2, G2> s) { return a.shr(s); }
    friend constexpr I operator<<(I a, U s) { return a.shl(s); }
    friend constexpr I operator>>(I a, U s) { return a.shr(s); }
    friend constexpr bool operator==(I a, I b) { return a.v == b.v; }
    friend constexpr bool operator!=(I a, I b) { return a.v != b.v; }
    friend constexpr bool operator<(I a, I b) { return a.v < b.v; }
    friend constexpr bool operator<=(I a, I b) { return a.v <= b.v; }
    friend constexpr bool operator>(I a, I b) { return a.v > b.v; }
    friend constexpr bool operator>=(I a, I b) { return a.v >= b.v; }
    constexpr I& operator+=(I b) { return *this = *this + b; }
    constexpr I& operator-=(I b) { return *this = *this - b; }
    constexpr I& operator*=(I b) { return *this = *this * b; }
    constexpr I& operator/=(I b) { return *this = *this / b; }
    constexpr I& operator%=(I b) { return *this = *this % b; }
    constexpr I& operator&=(I b) { return *this = *this & b; }
    constexpr I& operator|=(I b) { return *this = *this | b; }
    constexpr I& operator^=(I b) { return *this = *this ^ b; }
    template <class S> constexpr I& operator<<=(S s) { return *this = shl(s); }
    template <class S> constexpr I& operator>>=(S s) { return *this = shr(s); }
    constexpr I& operator++() { return *this = *this + from_raw(1); }
    constexpr I& operator--() { return *this = *this - from_raw(1); }
    constexpr I operator++(int) { I o = *this; ++*this; return o; }
    constexpr I operator--(int) { I o = *this; --*this; return o; }
};
// an untyped constant shifted by a variable takes the type the context gives it; the translated files only do this where that type is
// int or where the result is converted at once: computed in 128 bits, truncated by the conversion
template <class T, class G> constexpr U operator<<(U a, I<T, G> s) { return a << U((__int128)shift_count(s)); }
template <class T, class G> constexpr U operator>>(U a, I<T, G> s) { return a >> U((__int128)shift_count(s)); }

template <class X> struct is_int_wrapper : std::false_type {};
template <class T, class Tag> struct is_int_wrapper<I<T, Tag>> : std::true_type {};
using Int = I<int64_t>;   using Uint = I<uint64_t>;  using Uintptr = I<uint64_t>;
using Int8 = I<int8_t>;   using Int16 = I<int16_t>;  using Int32 = I<int32_t>;   using Int64 = I<int64_t>;
using Uint8 = I<uint8_t>; using Uint16 = I<uint16_t>; using Uint32 = I<uint32_t>; using Uint64 = I<uint64_t>;
using Byte = Uint8;       using Rune = Int32;
using Float64 = double;   using Float32 = float;
static_assert(sizeof(Byte) == 1 && sizeof(Int32) == 4 && sizeof(Uint64) == 8, "sized integers are their payload");

// `x := 5` gives an int, `x := 5.0` a float64; everything else keeps its type
constexpr Int def(U u) { return Int(u); }
template <class T> constexpr T def(T&& x) { return std::forward<T>(x); }
template <class T> constexpr T def(const T& x) { return x; }
struct UF { double v; };  // (untyped float constants are emitted as plain doubles)

// conversions T(x)
template <class To, class From> constexpr To conv(const From& x) {
    if constexpr (std::is_same_v<To, From>) return x;
    else if constexpr (std::is_floating_point_v<To> && std::is_same_v<From, U>) return (To)(double)x.v;
    else if constexpr (std::is_floating_point_v<To> && is_goint<From>::value) return (To)x.v;
    else if constexpr (std::is_pointer_v<To> && std::is_class_v<From> && !std::is_same_v<From, nil_t>) return (To)x.p;     // (*T)(unsafe.Pointer)
    else return To(x);
}
// array length / index from any integer
constexpr size_t csize(U u) { return (size_t)u.v; }
template <class T, class G> constexpr size_t csize(I<T, G> i) { return (size_t)i.v; }
constexpr int64_t idx64(U u) { return (int64_t)u.v; }
template <class T, class G> constexpr int64_t idx64(I<T, G> i) {
    if constexpr (std::is_same_v<T, uint64_t>) { if (i.v > (uint64_t)INT64_MAX) return -1; }
    return (int64_t)i.v;
}

// ---------------------------------------------------------------------------------------------------------------- arrays, slices
struct none_t {};
inline constexpr none_t none{};
[[noreturn]] inline void oob(int64_t i, int64_t n) { panic_str("runtime error: index out of range [" + std::to_string(i) + "] with length " + std::to_string(n)); }
[[noreturn]] inline void oob_slice(int64_t lo, int64_t hi, int64_t cap) {
    panic_str("runtime error: slice bounds out of range [" + std::to_string(lo) + ":" + std::to_string(hi) + "] with capacity " + std::to_string(cap));
}

template <class T> struct Slice {
    T* p = nullptr; int64_t n = 0, c = 0;
    Slice() = default;
    Slice(nil_t) {}
    Slice(T* p_, int64_t n_, int64_t c_) : p(p_), n(n_), c(c_) {}
    template <class K> T& operator[](K k) const { int64_t i = idx64(k); if ((uint64_t)i >= (uint64_t)n) oob(i, n); return p[i]; }
    static Slice make(int64_t n, int64_t c) {
        if (n < 0 || c < n) panic_str("makeslice: len out of range");
        T* q = (T*)alloc_zero((size_t)c * sizeof(T));
        // all-zero bytes ARE the zero value of integers, pointers, slices and of structs made of them; anything else (a string, a std::function, an
        // object with a vtable) is constructed in place, over the whole capacity (s[:cap(s)] may be read)
        // (never destructed: memory goes back with the call's arena, as Go's collector would take it; a std::string that outgrew its small buffer keeps
        // its heap block until the process ends - test infrastructure, bounded by what one call builds)
        if constexpr (!std::is_trivially_copyable_v<T>) for (int64_t i = 0; i < c; i++) new (q + i) T();
        return Slice(q, n, c);
    }
    static Slice lit(std::initializer_list<T> l) {
        Slice s = make((int64_t)l.size(), (int64_t)l.size());
        int64_t i = 0;
        for (const T& x : l) s.p[i++] = x;
        return s;
    }
    friend bool operator==(const Slice& s, nil_t) { return s.p == nullptr; }
    friend bool operator!=(const Slice& s, nil_t) { return s.p != nullptr; }
};

template <class T, size_t N> struct Array {
    T a[N ? N : 1];
    template <class K> T& operator[](K k) { int64_t i = idx64(k); if ((uint64_t)i >= (uint64_t)N) oob(i, (int64_t)N); return a[i]; }
    template <class K> const T& operator[](K k) const { int64_t i = idx64(k); if ((uint64_t)i >= (uint64_t)N) oob(i, (int64_t)N); return a[i]; }
    friend bool operator==(const Array& x, const Array& y) { for (size_t i = 0; i < N; i++) if (!(x.a[i] == y.a[i])) return false; return true; }
    friend bool operator!=(const Array& x, const Array& y) { return !(x == y); }
};

// ---------------------------------------------------------------------------------------------------------------- strings
struct String {
    std::string s;
    String() = default;
    String(const char* c) : s(c) {}
    String(const char* c, size_t n) : s(c, n) {}
    String(std::string x) : s(std::move(x)) {}
    template <class K> Byte operator[](K k) const { int64_t i = idx64(k); if ((uint64_t)i >= s.size()) oob(i, (int64_t)s.size()); return Byte::from_raw((uint8_t)s[(size_t)i]); }
    friend String operator+(const String& a, const String& b) { return String(a.s + b.s); }
    String& operator+=(const String& b) { s += b.s; return *this; }
    friend bool operator==(const String& a, const String& b) { return a.s == b.s; }
    friend bool operator!=(const String& a, const String& b) { return a.s != b.s; }
    friend bool operator<(const String& a, const String& b) { return a.s < b.s; }
    friend bool operator<=(const String& a, const String& b) { return a.s <= b.s; }
    friend bool operator>(const String& a, const String& b) { return a.s > b.s; }
    friend bool operator>=(const String& a, const String& b) { return a.s >= b.s; }
};
inline String operator""_s(const char* c, size_t n) { return String(c, n); }
constexpr String* dummy_string_ptr = nullptr;

// len / cap
template <class T> inline Int len(const Slice<T>& s) { return Int::from_raw(s.n); }
template <class T, size_t N> constexpr Int len(const Array<T, N>&) { return Int::from_raw((int64_t)N); }
inline Int len(const String& s) { return Int::from_raw((int64_t)s.s.size()); }
template <class T> inline Int cap(const Slice<T>& s) { return Int::from_raw(s.c); }
template <class T, size_t N> constexpr Int cap(const Array<T, N>&) { return Int::from_raw((int64_t)N); }

// slicing x[lo:hi] (bounds: 0 <= lo <= hi <= cap for slices, <= len for arrays and strings)
template <class K> inline int64_t bound(K k, int64_t) { return idx64(k); }
inline int64_t bound(none_t, int64_t d) { return d; }
template <class T, class L, class H> inline Slice<T> slice(const Slice<T>& s, L lo, H hi) {
    int64_t l = bound(lo, 0), h = bound(hi, s.n);
    if (l < 0 || h < l || h > s.c) oob_slice(l, h, s.c);
    return Slice<T>(s.p + l, h - l, s.c - l);
}
template <class T, class L, class H, class M> inline Slice<T> slice3(const Slice<T>& s, L lo, H hi, M mx) {
    int64_t l = bound(lo, 0), h = bound(hi, s.n), m = bound(mx, s.c);
    if (l < 0 || h < l || m < h || m > s.c) oob_slice(l, h, s.c);
    return Slice<T>(s.p + l, h - l, m - l);
}
template <class T, size_t N, class L, class H> inline Slice<T> slice(Array<T, N>& a, L lo, H hi) {
    int64_t l = bound(lo, 0), h = bound(hi, (int64_t)N);
    if (l < 0 || h < l || h > (int64_t)N) oob_slice(l, h, (int64_t)N);
    return Slice<T>(a.a + l, h - l, (int64_t)N - l);
}
template <class L, class H> inline String slice(const String& s, L lo, H hi) {
    int64_t l = bound(lo, 0), h = bound(hi, (int64_t)s.s.size());
    if (l < 0 || h < l || h > (int64_t)s.s.size()) oob_slice(l, h, (int64_t)s.s.size());
    return String(s.s.substr((size_t)l, (size_t)(h - l)));
}

// copy
template <class T> inline Int copy(const Slice<T>& d, const Slice<T>& s) {
    int64_t n = std::min(d.n, s.n);
    if constexpr (std::is_trivially_copyable_v<T>) { if (n > 0) std::memmove((void*)d.p, (const void*)s.p, (size_t)n * sizeof(T)); }
    else if (d.p <= s.p) { for (int64_t i = 0; i < n; i++) d.p[i] = s.p[i]; }
    else { for (int64_t i = n - 1; i >= 0; i--) d.p[i] = s.p[i]; }
    return Int::from_raw(n);
}
inline Int copy(const Slice<Byte>& d, const String& s) {
    int64_t n = std::min<int64_t>(d.n, (int64_t)s.s.size());
    if (n > 0) std::memcpy((void*)d.p, s.s.data(), (size_t)n);
    return Int::from_raw(n);
}

// make / new
template <class T, class N> inline Slice<T> make_slice(N n) { int64_t k = idx64(n); return Slice<T>::make(k, k); }
template <class T, class N, class C> inline Slice<T> make_slice(N n, C c) { return Slice<T>::make(idx64(n), idx64(c)); }
template <class T> inline T* New() { T* p = (T*)alloc_zero(sizeof(T)); return new (p) T(); }
template <class V> inline std::decay_t<V>* New(V&& v) { using T = std::decay_t<V>; T* p = (T*)alloc_zero(sizeof(T)); return new (p) T(std::forward<V>(v)); }

// append
template <class T> inline Slice<T> grow(const Slice<T>& s, int64_t need) {
    if (need <= s.c) return Slice<T>(s.p, s.n, s.c);
    int64_t nc = std::max<int64_t>(need, s.c < 256 ? 2 * s.c : s.c + s.c / 4 + 192);   // (capacity growth is not observable through len / contents)
    Slice<T> r = Slice<T>::make(s.n, nc);
    for (int64_t i = 0; i < s.n; i++) r.p[i] = s.p[i];
    return r;
}
template <class T> inline Slice<T> append(const Slice<T>& s) { return s; }
template <class T, class... A> inline Slice<T> append(const Slice<T>& s, const A&... xs) {
    Slice<T> r = grow(s, s.n + (int64_t)sizeof...(A));
    ((r.p[r.n++] = T(xs)), ...);
    return r;
}
template <class T> inline Slice<T> append_slice(const Slice<T>& s, const Slice<T>& t) {
    Slice<T> r = grow(s, s.n + t.n);
    if constexpr (std::is_trivially_copyable_v<T>) { if (t.n > 0) std::memmove((void*)(r.p + r.n), (const void*)t.p, (size_t)t.n * sizeof(T)); }
    else { for (int64_t i = 0; i < t.n; i++) r.p[r.n + i] = t.p[i]; }
    r.n += t.n;
    return r;
}
inline Slice<Byte> append_slice(const Slice<Byte>& s, const String& t) {
    Slice<Byte> r = grow(s, s.n + (int64_t)t.s.size());
    std::memcpy((void*)(r.p + r.n), t.s.data(), t.s.size());
    r.n += (int64_t)t.s.size();
    return r;
}

// clear
template <class T> inline void clear(const Slice<T>& s) { for (int64_t i = 0; i < s.n; i++) s.p[i] = T(); }

// string <-> []byte
inline Slice<Byte> bytes_of(const String& s) {
    Slice<Byte> r = Slice<Byte>::make((int64_t)s.s.size(), (int64_t)s.s.size());
    if (!s.s.empty()) std::memcpy((void*)r.p, s.s.data(), s.s.size());
    return r;
}
inline String string_of(const Slice<Byte>& b) { return String((const char*)b.p, (size_t)b.n); }
template <> inline Slice<Byte> conv<Slice<Byte>, String>(const String& s) { return bytes_of(s); }
template <> inline String conv<String, Slice<Byte>>(const Slice<Byte>& b) { return string_of(b); }

// range helpers: `for i := range x` over an integer, a slice, an array or a string (bytes; the translated files hold ASCII only there)
constexpr int64_t range_len(U u) { return (int64_t)u.v; }
template <class T, class G> constexpr int64_t range_len(I<T, G> i) { return (int64_t)i.v; }
template <class T> inline int64_t range_len(const Slice<T>& s) { return s.n; }
template <class T, size_t N> constexpr int64_t range_len(const Array<T, N>&) { return (int64_t)N; }
template <class T> inline T& range_at(const Slice<T>& s, int64_t i) { return s.p[i]; }
template <class T, size_t N> inline T& range_at(Array<T, N>& a, int64_t i) { return a.a[i]; }
template <class T, size_t N> inline const T& range_at(const Array<T, N>& a, int64_t i) { return a.a[i]; }
// (range over a pointer to an array: the Go idiom that avoids the copy)
template <class T, size_t N> constexpr int64_t range_len(Array<T, N>*) { return (int64_t)N; }
template <class T, size_t N> inline T& range_at(Array<T, N>* a, int64_t i) { return a->a[i]; }

// min / max builtins
template <class T> constexpr T min(T a) { return a; }
template <class T> constexpr T max(T a) { return a; }
template <class A, class B> constexpr auto min(A a, B b) {
    if constexpr (std::is_same_v<A, U> && !std::is_same_v<B, U>) return B(a) < b ? B(a) : b;
    else if constexpr (std::is_same_v<B, U> && !std::is_same_v<A, U>) return a < A(b) ? a : A(b);
    else return a < b ? a : b;
}
template <class A, class B> constexpr auto max(A a, B b) {
    if constexpr (std::is_same_v<A, U> && !std::is_same_v<B, U>) return B(a) > b ? B(a) : b;
    else if constexpr (std::is_same_v<B, U> && !std::is_same_v<A, U>) return a > A(b) ? a : A(b);
    else return a > b ? a : b;
}
template <class A, class B, class... R> constexpr auto min(A a, B b, R... r) { return min(min(a, b), r...); }
template <class A, class B, class... R> constexpr auto max(A a, B b, R... r) { return max(max(a, b), r...); }

// ---------------------------------------------------------------------------------------------------------------- error, any, map
struct error_iface { virtual String Error() = 0; virtual ~error_iface() = default; };
using error = error_iface*;
struct errorString : error_iface { String msg; explicit errorString(String m) : msg(std::move(m)) {} String Error() override { return msg; } };

template <class T> [[noreturn]] inline void panic(const T& v) {
    if constexpr (std::is_convertible_v<T, error>) { error e = v; throw PanicException(e ? e->Error().s : std::string("nil error"), true, (void*)e); }
    else if constexpr (std::is_same_v<T, String>) throw PanicException(v.s, false);
    else throw PanicException("panic", false);
}

// `any`: holds one value of a type the ctx maps of the translated constructors use
template <class K, class V> struct MapRef;
struct any;
template <class X> struct is_any_mapref : std::false_type {};
template <class K> struct is_any_mapref<MapRef<K, any>> : std::true_type {};
struct any {
    const std::type_info* ti = nullptr;
    std::shared_ptr<void> box;
    any() = default;
    any(nil_t) {}
    any(U u) : any(Int(u)) {}            // an untyped constant stored in an interface takes its default type
    template <class T, class = std::enable_if_t<!std::is_same_v<std::decay_t<T>, any> && !std::is_same_v<std::decay_t<T>, nil_t> && !std::is_same_v<std::decay_t<T>, U>>>
    any(T&& v) {
        using D = std::decay_t<T>;
        if constexpr (is_any_mapref<D>::value) { any e = v.operator any(); ti = e.ti; box = e.box; }                       // m[k] of a map of interfaces: the element, not the proxy (ctx["from"].(int))
        else if constexpr (std::is_convertible_v<D, error>) { ti = &typeid(error); box = std::make_shared<error>((error)v); }   // a value that implements `error` is kept as one
        else { ti = &typeid(D); box = std::make_shared<D>(std::forward<T>(v)); }
    }
    friend bool operator==(const any& a, nil_t) { return a.ti == nullptr; }
    friend bool operator!=(const any& a, nil_t) { return a.ti != nullptr; }
};
template <class T> inline bool type_is(const any& a) {
    if (!a.ti) return false;
    if (*a.ti == typeid(T)) return true;
    if constexpr (std::is_pointer_v<T> && std::is_convertible_v<T, error> && !std::is_same_v<T, error>) {
        if (*a.ti == typeid(error)) return dynamic_cast<T>(*(error*)a.box.get()) != nullptr;
    }
    return false;
}
template <class T> inline std::tuple<T, bool> assert2(const any& a) {
    if (a.ti && *a.ti == typeid(T)) return {*(T*)a.box.get(), true};
    if constexpr (std::is_pointer_v<T> && std::is_convertible_v<T, error> && !std::is_same_v<T, error>) {
        if (a.ti && *a.ti == typeid(error)) { T p = dynamic_cast<T>(*(error*)a.box.get()); if (p) return {p, true}; }
    }
    return {T(), false};
}
template <class T> inline T assert1(const any& a) {
    if (a.ti && *a.ti == typeid(T)) return *(T*)a.box.get();
    if constexpr (std::is_pointer_v<T> && std::is_convertible_v<T, error> && !std::is_same_v<T, error>) {
        if (a.ti && *a.ti == typeid(error)) { T p = dynamic_cast<T>(*(error*)a.box.get()); if (p) return p; }
    }
    panic_str(std::string("interface conversion: interface {} is ") + (a.ti ? a.ti->name() : "nil") + ", not " + typeid(T).name());
}
// x.(T) where x is a non-empty interface (a pointer to a polymorphic class here) and T a pointer type: a dynamic cast
template <class T, class I, class = std::enable_if_t<std::is_polymorphic_v<I> && std::is_pointer_v<T>>> inline T assert1(I* p) {
    T q = p ? dynamic_cast<T>(p) : nullptr;
    if (!q) panic_str(std::string("interface conversion: interface is ") + (p ? typeid(*p).name() : "nil") + ", not " + typeid(T).name());
    return q;
}
template <class T, class I, class = std::enable_if_t<std::is_polymorphic_v<I> && std::is_pointer_v<T>>> inline std::tuple<T, bool> assert2(I* p) {
    T q = p ? dynamic_cast<T>(p) : nullptr;
    return {q, q != nullptr};
}

// maps are references to shared storage, as in Go (a nil map reads as empty and panics on assignment)
template <class K, class V> struct Map;
// m[k] as a value reads WITHOUT inserting (a missing key is the zero value), m[k] = v inserts: a proxy tells the two apart
template <class K, class V> struct MapRef {
    Map<K, V>* owner; K key;
    operator V() const;
    MapRef& operator=(const V& v);
    MapRef& operator=(const MapRef& o) { return *this = (V)o; }
    template <class X, class = std::enable_if_t<std::is_convertible_v<X, V> && !std::is_same_v<std::decay_t<X>, V> && !std::is_same_v<std::decay_t<X>, MapRef>>>
    MapRef& operator=(X&& x) { return *this = V(std::forward<X>(x)); }
    auto operator->() const { return (V) * this; }
    friend bool operator==(const MapRef& r, nil_t) { return (V)r == nil; }
    friend bool operator!=(const MapRef& r, nil_t) { return (V)r != nil; }
};
template <class K, class V> struct Map {
    std::shared_ptr<std::map<K, V>> m;
    Map() = default;
    Map(nil_t) {}
    MapRef<K, V> operator[](const K& k) { return MapRef<K, V>{this, k}; }
    friend bool operator==(const Map& a, nil_t) { return !a.m; }
    friend bool operator!=(const Map& a, nil_t) { return (bool)a.m; }
};
template <class K, class V> MapRef<K, V>::operator V() const { if (!owner->m) return V(); auto it = owner->m->find(key); return it == owner->m->end() ? V() : it->second; }
template <class K, class V> MapRef<K, V>& MapRef<K, V>::operator=(const V& v) { if (!owner->m) panic_str("assignment to entry in nil map"); (*owner->m)[key] = v; return *this; }
template <class K, class V> inline V def(MapRef<K, V> r) { return (V)r; }
template <class K, class V> inline std::tuple<V, bool> map_get2(const Map<K, V>& m, const K& k) {
    if (!m.m) return {V(), false};
    auto it = m.m->find(k);
    if (it == m.m->end()) return {V(), false};
    return {it->second, true};
}
template <class K, class V> inline Int len(const Map<K, V>& m) { return Int::from_raw(m.m ? (int64_t)m.m->size() : 0); }
template <class K, class V> inline Map<K, V> make_map() { Map<K, V> r; r.m = std::make_shared<std::map<K, V>>(); return r; }
template <class K, class V> inline void map_delete(Map<K, V>& m, const K& k) { if (m.m) m.m->erase(k); }

// ---------------------------------------------------------------------------------------------------------------- defer / recover
struct DeferFrame {
    std::vector<std::function<void()>> fns;
    bool panicking = false, recovered = false, is_error = true;
    void* err_obj = nullptr;
    std::string msg;
    DeferFrame* prev = nullptr;
    template <class F> void push(F&& f) { fns.emplace_back(std::forward<F>(f)); }
    void set_panic(const PanicException& e) { panicking = true; recovered = false; msg = e.what(); is_error = e.is_error; err_obj = e.err_obj; }
    void run();
};
inline thread_local DeferFrame* g_running_defers = nullptr;
inline void DeferFrame::run() {
    prev = g_running_defers;
    g_running_defers = this;
    while (!fns.empty()) {
        std::function<void()> f = std::move(fns.back());
        fns.pop_back();
        try { f(); } catch (const PanicException& e) { set_panic(e); }        // a panic inside a deferred call replaces the current one
    }
    g_running_defers = prev;
    if (panicking && !recovered) throw PanicException(msg, is_error, err_obj);
}
// recover(): the value of the panic that is unwinding through the function whose deferred call this is, or nil
inline any recover() {
    DeferFrame* f = g_running_defers;
    if (f == nullptr || !f->panicking || f->recovered) return any();
    f->recovered = true;
    if (f->is_error && f->err_obj) return any((error)f->err_obj);
    if (f->is_error) return any((error)New<errorString>(errorString(String(f->msg))));
    return any(String(f->msg));
}

// ---------------------------------------------------------------------------------------------------------------- for range
template <class R> struct Ranger;
template <class T> struct Ranger<Slice<T>> { const Slice<T>& s; int64_t n; Int key(int64_t i) const { return Int::from_raw(i); } T& val(int64_t i) const { return s.p[i]; } };
template <class T, size_t N> struct Ranger<Array<T, N>> { Array<T, N> a; int64_t n; Int key(int64_t i) const { return Int::from_raw(i); } T val(int64_t i) const { return a.a[i]; } };   // (range over an array VALUE iterates a copy)
template <class T, size_t N> struct Ranger<Array<T, N>*> { Array<T, N>* a; int64_t n; Int key(int64_t i) const { return Int::from_raw(i); } T& val(int64_t i) const { return a->a[i]; } };
template <class K, class V> struct Ranger<Map<K, V>> {
    std::vector<std::pair<K, V>> items; int64_t n;
    K key(int64_t i) const { return items[(size_t)i].first; }
    V val(int64_t i) const { return items[(size_t)i].second; }
};
struct IntRanger { int64_t n; Int key(int64_t i) const { return Int::from_raw(i); } };
struct StringRanger {          // (bytes; the translated files range over ASCII strings only)
    const String& s; int64_t n;
    Int key(int64_t i) const { return Int::from_raw(i); }
    Rune val(int64_t i) const { const unsigned char c = (unsigned char)s.s[(size_t)i]; if (c >= 0x80) panic_str("go2cpp: range over a non-ASCII string is not translated"); return Rune::from_raw((int32_t)c); }
};
template <class T> inline Ranger<Slice<T>> ranger(const Slice<T>& s) { return {s, s.n}; }
template <class T, size_t N> inline Ranger<Array<T, N>> ranger(const Array<T, N>& a) { return {a, (int64_t)N}; }
template <class T, size_t N> inline Ranger<Array<T, N>*> ranger(Array<T, N>* a) { return {a, (int64_t)N}; }
template <class K, class V> inline Ranger<Map<K, V>> ranger(const Map<K, V>& m) {
    Ranger<Map<K, V>> r;
    if (m.m) for (auto& kv : *m.m) r.items.emplace_back(kv.first, kv.second);
    r.n = (int64_t)r.items.size();
    return r;
}
// `for i := range x` (no value variable): the range expression is only measured, never copied
template <class X> inline auto ranger_keys(const X& x) { return ranger(x); }
template <class T, size_t N> inline auto ranger_keys(const Array<T, N>&);
inline IntRanger ranger(U u) { return {(int64_t)u.v}; }
template <class T, class G> inline IntRanger ranger(I<T, G> i) { return {(int64_t)i.v}; }
inline StringRanger ranger(const String& s) { return {s, (int64_t)s.s.size()}; }
template <class T, size_t N> inline auto ranger_keys(const Array<T, N>&) { return IntRanger{(int64_t)N}; }

}  // namespace go

// ---------------------------------------------------------------------------------------------------------------- std packages
namespace go_errors {
inline go::error New(const go::String& s) { return go::New<go::errorString>(go::errorString(s)); }
inline bool Is(go::error err, go::error target) { return err == target; }     // (no translated error type wraps another)
}

namespace go_fmt {
inline void fmt_arg(std::string& out, const go::String& v) { out += v.s; }
inline void fmt_arg(std::string& out, const char* v) { out += v; }
inline void fmt_arg(std::string& out, go::U v) { out += std::to_string((long long)v.v); }
inline void fmt_arg(std::string& out, bool v) { out += v ? "true" : "false"; }
inline void fmt_arg(std::string& out, double v) { out += std::to_string(v); }
inline void fmt_arg(std::string& out, go::error e) { out += e ? e->Error().s : "<nil>"; }
template <class T, class G> inline void fmt_arg(std::string& out, go::I<T, G> v) {
    if constexpr (std::is_signed_v<T>) out += std::to_string((long long)v.v); else out += std::to_string((unsigned long long)v.v);
}
template <class T> inline void fmt_arg(std::string& out, const T&) { out += "?"; }
// one argument of a formatting call: its default rendering (%v) and, for integers, sign and magnitude (for %d %b %o %x %X %c with flags, width, precision)
struct FmtArg { std::string text; bool is_int = false, neg = false; unsigned long long mag = 0; };
template <class T> inline FmtArg fmt_capture(const T& v) {
    FmtArg a; fmt_arg(a.text, v);
    if constexpr (go::is_int_wrapper<T>::value) {
        a.is_int = true;
        if constexpr (std::is_signed_v<decltype(v.v)>) { a.neg = v.v < 0; a.mag = a.neg ? 0ull - (unsigned long long)v.v : (unsigned long long)v.v; }
        else a.mag = (unsigned long long)v.v;
    } else if constexpr (std::is_same_v<T, go::U>) { a.is_int = true; a.neg = v.v < 0; a.mag = (unsigned long long)(a.neg ? -v.v : v.v); }
    return a;
}
inline std::string fmt_digits(unsigned long long m, unsigned base, bool upper) {
    if (m == 0) return "0";
    std::string d;
    while (m) { unsigned k = (unsigned)(m % base); d += (char)(k < 10 ? '0' + k : (upper ? 'A' : 'a') + (k - 10)); m /= base; }
    return std::string(d.rbegin(), d.rend());
}
// fmt.Sprintf for the verbs the translated files use: %v %d %s %q %x %X %b %o %c with the flags - 0 +, a width and a precision (for integers a precision
// is the least number of digits, as in Go); anything else renders the argument's default form
template <class... A> inline go::String Sprintf(const go::String& f, const A&... a) {
    std::vector<FmtArg> args;
    (args.push_back(fmt_capture(a)), ...);
    std::string out;
    size_t k = 0;
    for (size_t i = 0; i < f.s.size(); i++) {
        if (f.s[i] != '%') { out += f.s[i]; continue; }
        if (i + 1 < f.s.size() && f.s[i + 1] == '%') { out += '%'; i++; continue; }
        size_t j = i + 1;
        bool left = false, zero = false, plus = false;
        for (; j < f.s.size() && (f.s[j] == '-' || f.s[j] == '0' || f.s[j] == '+' || f.s[j] == ' ' || f.s[j] == '#'); j++) { left |= f.s[j] == '-'; zero |= f.s[j] == '0'; plus |= f.s[j] == '+'; }
        int width = -1, prec = -1;
        if (j < f.s.size() && isdigit((unsigned char)f.s[j])) { width = 0; while (j < f.s.size() && isdigit((unsigned char)f.s[j])) width = width * 10 + (f.s[j++] - '0'); }
        if (j < f.s.size() && f.s[j] == '.') { j++; prec = 0; while (j < f.s.size() && isdigit((unsigned char)f.s[j])) prec = prec * 10 + (f.s[j++] - '0'); }
        const char verb = j < f.s.size() ? f.s[j] : 'v';
        i = j;
        if (k >= args.size()) { out += "%!"; out += verb; out += "(MISSING)"; continue; }
        const FmtArg& g = args[k++];
        std::string body;
        if (g.is_int && (verb == 'd' || verb == 'v' || verb == 'b' || verb == 'o' || verb == 'x' || verb == 'X')) {
            std::string d = fmt_digits(g.mag, verb == 'b' ? 2 : verb == 'o' ? 8 : (verb == 'x' || verb == 'X') ? 16 : 10, verb == 'X');
            if (prec >= 0 && (int)d.size() < prec) d = std::string((size_t)prec - d.size(), '0') + d;
            const std::string sign = g.neg ? "-" : (plus ? "+" : "");
            if (zero && !left && prec < 0 && width > (int)(d.size() + sign.size())) d = std::string((size_t)width - d.size() - sign.size(), '0') + d;
            body = sign + d;
        } else if (g.is_int && verb == 'c') body = std::string(1, (char)g.mag);
        else if (verb == 'q') body = "\"" + g.text + "\"";
        else { body = g.text; if (prec >= 0 && !g.is_int && (int)body.size() > prec) body.resize((size_t)prec); }
        if (width > (int)body.size()) body = left ? body + std::string((size_t)width - body.size(), ' ') : std::string((size_t)width - body.size(), ' ') + body;
        out += body;
    }
    return go::String(out);
}
template <class... A> inline go::error Errorf(const go::String& f, const A&... a) { return go_errors::New(Sprintf(f, a...)); }
template <class... A> inline void Printf(const go::String& f, const A&... a) { std::fputs(Sprintf(f, a...).s.c_str(), stderr); }
template <class... A> inline void Println(const A&... a) { std::string s; ((fmt_arg(s, a), s += ' '), ...); s += '\n'; std::fputs(s.c_str(), stderr); }
template <class... A> inline go::String Sprint(const A&... a) { std::string s; (fmt_arg(s, a), ...); return go::String(s); }
}  // namespace go_fmt

namespace go_binary {
struct BigEndian_t {
    BigEndian_t* operator->() { return this; }
    const BigEndian_t* operator->() const { return this; }
    static void need(const go::Slice<go::Byte>& b, int64_t n) { if (b.n < n) go::oob(n - 1, b.n); }
    go::Uint16 Uint16(const go::Slice<go::Byte>& b) const { need(b, 2); return go::Uint16::from_raw((uint16_t)((b.p[0].v << 8) | b.p[1].v)); }
    go::Uint32 Uint32(const go::Slice<go::Byte>& b) const { need(b, 4); uint32_t x; std::memcpy(&x, b.p, 4); return go::Uint32::from_raw(__builtin_bswap32(x)); }
    go::Uint64 Uint64(const go::Slice<go::Byte>& b) const { need(b, 8); uint64_t x; std::memcpy(&x, b.p, 8); return go::Uint64::from_raw(__builtin_bswap64(x)); }
    void PutUint16(const go::Slice<go::Byte>& b, go::Uint16 v) const { need(b, 2); b.p[0].v = (uint8_t)(v.v >> 8); b.p[1].v = (uint8_t)v.v; }
    void PutUint32(const go::Slice<go::Byte>& b, go::Uint32 v) const { need(b, 4); uint32_t x = __builtin_bswap32(v.v); std::memcpy(b.p, &x, 4); }
    void PutUint64(const go::Slice<go::Byte>& b, go::Uint64 v) const { need(b, 8); uint64_t x = __builtin_bswap64(v.v); std::memcpy(b.p, &x, 8); }
};
struct LittleEndian_t {
    LittleEndian_t* operator->() { return this; }
    const LittleEndian_t* operator->() const { return this; }
    static void need(const go::Slice<go::Byte>& b, int64_t n) { if (b.n < n) go::oob(n - 1, b.n); }
    go::Uint16 Uint16(const go::Slice<go::Byte>& b) const { need(b, 2); uint16_t x; std::memcpy(&x, b.p, 2); return go::Uint16::from_raw(x); }
    go::Uint32 Uint32(const go::Slice<go::Byte>& b) const { need(b, 4); uint32_t x; std::memcpy(&x, b.p, 4); return go::Uint32::from_raw(x); }
    go::Uint64 Uint64(const go::Slice<go::Byte>& b) const { need(b, 8); uint64_t x; std::memcpy(&x, b.p, 8); return go::Uint64::from_raw(x); }
    void PutUint16(const go::Slice<go::Byte>& b, go::Uint16 v) const { need(b, 2); std::memcpy(b.p, &v.v, 2); }
    void PutUint32(const go::Slice<go::Byte>& b, go::Uint32 v) const { need(b, 4); std::memcpy(b.p, &v.v, 4); }
    void PutUint64(const go::Slice<go::Byte>& b, go::Uint64 v) const { need(b, 8); std::memcpy(b.p, &v.v, 8); }
};
inline BigEndian_t BigEndian;
inline LittleEndian_t LittleEndian;
}  // namespace go_binary

namespace go_bits {
inline go::Int TrailingZeros64(go::Uint64 x) { return go::Int::from_raw(x.v ? __builtin_ctzll(x.v) : 64); }
inline go::Int TrailingZeros32(go::Uint32 x) { return go::Int::from_raw(x.v ? __builtin_ctz(x.v) : 32); }
inline go::Int LeadingZeros64(go::Uint64 x) { return go::Int::from_raw(x.v ? __builtin_clzll(x.v) : 64); }
inline go::Int LeadingZeros32(go::Uint32 x) { return go::Int::from_raw(x.v ? __builtin_clz(x.v) : 32); }
inline go::Int Len64(go::Uint64 x) { return go::Int::from_raw(x.v ? 64 - __builtin_clzll(x.v) : 0); }
inline go::Int Len32(go::Uint32 x) { return go::Int::from_raw(x.v ? 32 - __builtin_clz(x.v) : 0); }
inline go::Int OnesCount64(go::Uint64 x) { return go::Int::from_raw(__builtin_popcountll(x.v)); }
inline go::Uint32 RotateLeft32(go::Uint32 x, go::Int k) { unsigned s = (unsigned)k.v & 31; return go::Uint32::from_raw(s ? (x.v << s) | (x.v >> (32 - s)) : x.v); }
inline go::Uint64 RotateLeft64(go::Uint64 x, go::Int k) { unsigned s = (unsigned)k.v & 63; return go::Uint64::from_raw(s ? (x.v << s) | (x.v >> (64 - s)) : x.v); }
inline go::Uint64 ReverseBytes64(go::Uint64 x) { return go::Uint64::from_raw(__builtin_bswap64(x.v)); }
inline go::Uint32 ReverseBytes32(go::Uint32 x) { return go::Uint32::from_raw(__builtin_bswap32(x.v)); }
}  // namespace go_bits

namespace go_sort {
inline void Ints(const go::Slice<go::Int>& s) { std::sort(s.p, s.p + s.n, [](go::Int a, go::Int b) { return a.v < b.v; }); }
}
namespace go_slices {
template <class T, class F> inline void SortStableFunc(const go::Slice<T>& s, F cmp) {
    std::stable_sort(s.p, s.p + s.n, [&](const T& a, const T& b) { return cmp(a, b) < go::Int(); });
}
template <class T, class F> inline void SortFunc(const go::Slice<T>& s, F cmp) {
    std::sort(s.p, s.p + s.n, [&](const T& a, const T& b) { return cmp(a, b) < go::Int(); });
}
}
namespace go_io {
struct Reader { virtual std::tuple<go::Int, go::error> Read(go::Slice<go::Byte> b) = 0; virtual ~Reader() = default; };
struct Writer { virtual std::tuple<go::Int, go::error> Write(go::Slice<go::Byte> b) = 0; virtual ~Writer() = default; };
struct Closer { virtual go::error Close() = 0; virtual ~Closer() = default; };
// (an interface that embeds others converts to them: virtual bases)
struct ReadCloser : virtual Reader, virtual Closer {};
struct WriteCloser : virtual Writer, virtual Closer {};
struct ReadWriteCloser : virtual ReadCloser, virtual WriteCloser {};
inline go::errorString EOF_value{go::String("EOF")};
inline go::error EOF_ = &EOF_value;
struct nopCloser : virtual ReadCloser {
    Reader* r;
    explicit nopCloser(Reader* r_) : r(r_) {}
    std::tuple<go::Int, go::error> Read(go::Slice<go::Byte> b) override { return r->Read(b); }
    go::error Close() override { return nullptr; }
};
inline ReadCloser* NopCloser(Reader* r) { return go::New<nopCloser>(nopCloser(r)); }
}  // namespace go_io
namespace go_fmt {
template <class... A> inline std::tuple<go::Int, go::error> Fprintf(go_io::Writer* w, const go::String& f, const A&... a) {
    go::String s = Sprintf(f, a...);
    go::Slice<go::Byte> b = go::Slice<go::Byte>::make((int64_t)s.s.size(), (int64_t)s.s.size());
    if (!s.s.empty()) std::memcpy((void*)b.p, s.s.data(), s.s.size());
    return w->Write(b);
}
}  // namespace go_fmt
namespace go_os {
// os.Stdout / os.Stderr as io.Writers (the debug bit streams print through them): both go to stderr here, stdout belongs to the caller
struct File : virtual go_io::ReadWriteCloser {
    File* operator->() { return this; }
    FILE* f = nullptr;               // nullptr: the process's stderr
    std::string path;
    std::tuple<go::Int, go::error> Write(go::Slice<go::Byte> b) override {
        size_t n = b.n ? std::fwrite((const void*)b.p, 1, (size_t)b.n, f ? f : stderr) : 0;
        if ((int64_t)n != b.n) return {go::Int::from_raw((int64_t)n), go_errors::New(go::String("write " + path + ": short write"))};
        return {go::Int::from_raw(b.n), nullptr};
    }
    std::tuple<go::Int, go::error> Read(go::Slice<go::Byte> b) override {
        if (!f) return {go::Int(), go_errors::New(go::String("read: file is not open"))};
        size_t n = b.n ? std::fread((void*)b.p, 1, (size_t)b.n, f) : 0;
        if (n == 0 && b.n > 0) return {go::Int(), go_io::EOF_};
        return {go::Int::from_raw((int64_t)n), nullptr};
    }
    go::error Close() override { if (f) { std::fclose(f); f = nullptr; } return nullptr; }
    go::String Name() { return go::String(path); }
};
inline File stdout_file, stderr_file;
inline File* Stdout = &stdout_file;
inline File* Stderr = &stderr_file;
// os.CreateTemp(dir, pattern): a new file whose name replaces the last * of pattern by a random string
inline std::tuple<File*, go::error> CreateTemp(const go::String& dir, const go::String& pattern) {
    std::string d = dir.s.empty() ? std::string(std::getenv("TMPDIR") ? std::getenv("TMPDIR") : "/tmp") : dir.s;
    std::string pre = pattern.s, suf;
    size_t star = pattern.s.rfind('*');
    if (star != std::string::npos) { pre = pattern.s.substr(0, star); suf = pattern.s.substr(star + 1); }
    std::string tmpl = d + "/" + pre + "XXXXXX" + suf;
    std::vector<char> buf(tmpl.begin(), tmpl.end()); buf.push_back(0);
    int fd = mkstemps(buf.data(), (int)suf.size());
    if (fd < 0) return {nullptr, go_errors::New(go::String("CreateTemp: cannot create " + tmpl))};
    File* r = go::New<File>();
    r->f = fdopen(fd, "w+b");
    r->path = buf.data();
    return {r, nullptr};
}
inline std::tuple<File*, go::error> Open(const go::String& name) {
    FILE* f = std::fopen(name.s.c_str(), "rb");
    if (!f) return {nullptr, go_errors::New(go::String("open " + name.s + ": no such file or directory"))};
    File* r = go::New<File>();
    r->f = f; r->path = name.s;
    return {r, nullptr};
}
inline go::error Remove(const go::String& name) { return std::remove(name.s.c_str()) == 0 ? nullptr : go_errors::New(go::String("remove " + name.s + ": failed")); }
}  // namespace go_os
namespace go_sync {
// goroutines of the translated files run one after the other (`go f(x)` is emitted as the call): a WaitGroup has nothing to wait for
struct WaitGroup {
    WaitGroup* operator->() { return this; }
    template <class T> void Add(T) {}
    void Done() {}
    void Wait() {}
};
struct Mutex { Mutex* operator->() { return this; } void Lock() {} void Unlock() {} };
}  // namespace go_sync

namespace go_strings {
inline go::String ToUpper(const go::String& s) { std::string r = s.s; for (char& c : r) c = (char)toupper((unsigned char)c); return go::String(r); }
inline go::String ToLower(const go::String& s) { std::string r = s.s; for (char& c : r) c = (char)tolower((unsigned char)c); return go::String(r); }
inline go::Int IndexByte(const go::String& s, go::Byte c) { size_t k = s.s.find((char)c.v); return go::Int::from_raw(k == std::string::npos ? -1 : (int64_t)k); }
inline bool Contains(const go::String& s, const go::String& sub) { return s.s.find(sub.s) != std::string::npos; }
inline bool HasPrefix(const go::String& s, const go::String& pre) { return s.s.compare(0, pre.s.size(), pre.s) == 0; }
inline go::Slice<go::String> Split(const go::String& s, const go::String& sep) {
    std::vector<go::String> parts;
    size_t a = 0;
    while (true) {
        size_t k = sep.s.empty() ? std::string::npos : s.s.find(sep.s, a);
        if (k == std::string::npos) { parts.push_back(go::String(s.s.substr(a))); break; }
        parts.push_back(go::String(s.s.substr(a, k - a)));
        a = k + sep.s.size();
    }
    // (String is not trivially copyable: the slice's cells are constructed in place and never destroyed -- arena memory)
    go::String* mem = (go::String*)go::alloc_zero(parts.size() * sizeof(go::String));
    for (size_t i = 0; i < parts.size(); i++) new (mem + i) go::String(parts[i]);
    return go::Slice<go::String>(mem, (int64_t)parts.size(), (int64_t)parts.size());
}
inline go::String Repeat(const go::String& s, go::Int count) {
    if (count.v < 0) go::panic_str("strings: negative Repeat count");
    std::string r; r.reserve(s.s.size() * (size_t)count.v);
    for (int64_t i = 0; i < count.v; i++) r += s.s;
    return go::String(r);
}
}  // namespace go_strings

namespace go_bytes {
// bytes.Buffer as internal.BufferStream uses it. NewBuffer(b) takes OWNERSHIP of b's backing array: writes append in place while they fit into
// cap(b) (the block task of io/CompressedStream.go relies on that: it reads the encoded bits back through its own slice of the same array),
// then the buffer moves to a larger array; reads consume from the front.
struct Buffer : virtual go_io::Writer, virtual go_io::Reader {
    Buffer* operator->() { return this; }
    go::Slice<go::Byte> buf;
    int64_t off = 0;
    go::String String() { return go::String(std::string((const char*)buf.p + off, (size_t)(buf.n - off))); }
    std::tuple<go::Int, go::error> Write(go::Slice<go::Byte> b) override {
        if (buf.n + b.n > buf.c) {
            int64_t nc = std::max<int64_t>(2 * buf.c + b.n, 64);
            go::Slice<go::Byte> nb = go::Slice<go::Byte>::make(buf.n, nc);
            if (buf.n) std::memcpy((void*)nb.p, (const void*)buf.p, (size_t)buf.n);
            buf = nb;
        }
        if (b.n) std::memmove((void*)(buf.p + buf.n), (const void*)b.p, (size_t)b.n);
        buf.n += b.n;
        return {go::Int::from_raw(b.n), nullptr};
    }
    std::tuple<go::Int, go::error> Read(go::Slice<go::Byte> b) override {
        int64_t n = std::min<int64_t>(b.n, buf.n - off);
        if (n == 0 && b.n > 0) return {go::Int(), go_io::EOF_};
        if (n) std::memmove((void*)b.p, (const void*)(buf.p + off), (size_t)n);
        off += n;
        return {go::Int::from_raw(n), nullptr};
    }
    go::Int Len() { return go::Int::from_raw(buf.n - off); }
    go::Int Available() { return go::Int::from_raw(buf.c - buf.n); }
};
struct Reader : virtual go_io::Reader {
    Reader* operator->() { return this; }
    go::Slice<go::Byte> b;
    int64_t off = 0;
    std::tuple<go::Int, go::error> Read(go::Slice<go::Byte> d) override {
        if (off >= b.n) return {go::Int(), go_io::EOF_};
        int64_t n = std::min<int64_t>(d.n, b.n - off);
        if (n) std::memmove((void*)d.p, (const void*)(b.p + off), (size_t)n);
        off += n;
        return {go::Int::from_raw(n), nullptr};
    }
    go::Int Len() { return go::Int::from_raw(b.n - off); }
};
inline Reader* NewReader(go::Slice<go::Byte> b) { Reader* r = go::New<Reader>(); r->b = b; return r; }
inline Buffer* NewBuffer(go::Slice<go::Byte> b) { Buffer* r = go::New<Buffer>(); r->buf = b; return r; }
inline bool Equal(go::Slice<go::Byte> a, go::Slice<go::Byte> b) { return a.n == b.n && (a.n == 0 || std::memcmp((const void*)a.p, (const void*)b.p, (size_t)a.n) == 0); }
inline go::Slice<go::Byte> Repeat(go::Slice<go::Byte> b, go::Int count) {
    if (count.v < 0) go::panic_str("bytes: negative Repeat count");
    go::Slice<go::Byte> r = go::Slice<go::Byte>::make(b.n * count.v, b.n * count.v);
    for (int64_t i = 0; i < count.v; i++) if (b.n) std::memcpy((void*)(r.p + i * b.n), (const void*)b.p, (size_t)b.n);
    return r;
}
}  // namespace go_bytes

namespace go_atomic {
// the goroutines of the translated files run one after the other: plain loads and stores
template <class T> inline T LoadInt32(T* p) { return *p; }
template <class T, class V> inline void StoreInt32(T* p, V v) { *p = T(v); }
template <class T, class V> inline T SwapInt32(T* p, V v) { T o = *p; *p = T(v); return o; }
template <class T, class V> inline T AddInt32(T* p, V v) { *p = *p + T(v); return *p; }
template <class T, class A, class B> inline bool CompareAndSwapInt32(T* p, A o, B n) { if (*p == T(o)) { *p = T(n); return true; } return false; }
template <class T> inline T LoadInt64(T* p) { return *p; }
template <class T, class V> inline void StoreInt64(T* p, V v) { *p = T(v); }
template <class T, class V> inline T AddInt64(T* p, V v) { *p = *p + T(v); return *p; }
template <class T> inline T LoadUint64(T* p) { return *p; }
template <class T, class V> inline void StoreUint64(T* p, V v) { *p = T(v); }
template <class T, class V> inline T AddUint64(T* p, V v) { *p = *p + T(v); return *p; }
}  // namespace go_atomic
namespace go_time {
struct Duration_tag {};
using Duration = go::I<int64_t, Duration_tag>;
struct Time {
    Time* operator->() { return this; }
    const Time* operator->() const { return this; }
    int64_t ns = 0;
    bool IsZero() const { return ns == 0; }
    go::Int64 UnixNano() const { return go::Int64::from_raw(ns); }
    go::Int64 UnixMilli() const { return go::Int64::from_raw(ns / 1000000); }
    Duration Sub(const Time& o) const { return Duration::from_raw(ns - o.ns); }
    bool Before(const Time& o) const { return ns < o.ns; }
    bool After(const Time& o) const { return ns > o.ns; }
};
inline Time Now() { timespec ts; clock_gettime(CLOCK_REALTIME, &ts); Time t; t.ns = (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec; return t; }
inline Duration Since(const Time& t) { return Now().Sub(t); }
constexpr Duration Nanosecond = Duration::from_raw(1), Microsecond = Duration::from_raw(1000), Millisecond = Duration::from_raw(1000000), Second = Duration::from_raw(1000000000);
}  // namespace go_time
namespace go_runtime {
inline void Gosched() {}
struct Pinner { Pinner* operator->() { return this; } template <class T> void Pin(T*) {} void Unpin() {} };   // (memory does not move here)
template <class T, class F> inline void SetFinalizer(T*, F) {}
inline go::Int NumCPU() { return go::Int::from_raw(1); }
}  // namespace go_runtime

// ---- what the reference's own *_test.go files need (they are translated and run as tests of the translation, and on the device) ----
namespace go_testing {
struct FailNow {};                                   // t.Fatal / t.FailNow: leaves the test function
struct T {
    T* operator->() { return this; }
    std::string name, log;
    bool failed = false;
    template <class... A> void note(const A&... a) { std::string s; ((go_fmt::fmt_arg(s, a), s += ' '), ...); log += name + ": " + s + "\n"; }
    template <class... A> void Log(const A&... a) { note(a...); }
    template <class... A> void Logf(const go::String& f, const A&... a) { note(go_fmt::Sprintf(f, a...)); }
    template <class... A> void Error(const A&... a) { failed = true; note(a...); }
    template <class... A> void Errorf(const go::String& f, const A&... a) { failed = true; note(go_fmt::Sprintf(f, a...)); }
    template <class... A> [[noreturn]] void Fatal(const A&... a) { failed = true; note(a...); throw FailNow{}; }
    template <class... A> [[noreturn]] void Fatalf(const go::String& f, const A&... a) { failed = true; note(go_fmt::Sprintf(f, a...)); throw FailNow{}; }
    [[noreturn]] void FailNow_() { failed = true; throw FailNow{}; }
    void Fail() { failed = true; }
    bool Failed() const { return failed; }
    void Helper() {}
    void Parallel() {}
    go::String Name() const { return go::String(name); }
    template <class F> bool Run(const go::String& sub, F f) {
        T t; t.name = name + "/" + sub.s;
        try { f(&t); } catch (FailNow&) {}
        if (t.failed) failed = true;
        log += t.log;
        return !t.failed;
    }
};
inline bool Verbose() { return false; }
inline bool Short() { return false; }
using TestFn = void (*)(T*);
inline std::vector<std::pair<std::string, TestFn>>& registry() { static std::vector<std::pair<std::string, TestFn>> r; return r; }
struct Register { Register(const char* pkg, const char* name, TestFn f) { registry().emplace_back(std::string(pkg) + "." + name, f); } };
// runs one registered test: 0 = pass, 1 = fail, 2 = no such test; the log of t.Log / t.Error / a panic text goes to `log`
inline int run(const std::string& full, std::string& log) {
    for (auto& e : registry()) {
        if (e.first != full) continue;
        T t; t.name = full;
        try { e.second(&t); }
        catch (FailNow&) {}
        catch (go::PanicException& p) { t.failed = true; t.log += full + ": panic: " + p.what() + "\n"; }
        catch (std::exception& x) { t.failed = true; t.log += full + ": exception: " + x.what() + "\n"; }
        log = t.log;
        return t.failed ? 1 : 0;
    }
    return 2;
}
}  // namespace go_testing
namespace go_rand {
// math/rand: the tests draw inputs from it, never compare against Go's own sequence (they are round trips), so any generator serves. Seeded per
// process by the runner (KREF_TEST_SEED) so that a failure can be replayed.
inline uint64_t& state() { static uint64_t s = 0x9E3779B97F4A7C15ull; return s; }
inline uint64_t next(uint64_t& s) { s += 0x9E3779B97F4A7C15ull; uint64_t z = s; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
struct Source { uint64_t s; };
struct Rand {
    Rand* operator->() { return this; }
    uint64_t s = 1;
    go::Int Intn(go::Int n) { if (n.v <= 0) go::panic_str("invalid argument to Intn"); return go::Int::from_raw((int64_t)(next(s) % (uint64_t)n.v)); }
    go::Int Int() { return go::Int::from_raw((int64_t)(next(s) >> 1)); }
    go::Int32 Int31n(go::Int32 n) { if (n.v <= 0) go::panic_str("invalid argument to Int31n"); return go::Int32::from_raw((int32_t)(next(s) % (uint64_t)n.v)); }
    go::Uint32 Uint32() { return go::Uint32::from_raw((uint32_t)next(s)); }
    go::Uint64 Uint64() { return go::Uint64::from_raw(next(s)); }
};
inline Source NewSource(go::Int64 seed) { return Source{(uint64_t)seed.v ^ state()}; }
inline Rand* New(Source src) { Rand* r = go::New<Rand>(); r->s = src.s; return r; }
inline go::Int Intn(go::Int n) { if (n.v <= 0) go::panic_str("invalid argument to Intn"); return go::Int::from_raw((int64_t)(next(state()) % (uint64_t)n.v)); }
inline go::Int Int() { return go::Int::from_raw((int64_t)(next(state()) >> 1)); }
inline go::Int32 Int31n(go::Int32 n) { return go::Int32::from_raw((int32_t)(next(state()) % (uint64_t)n.v)); }
inline go::Uint32 Uint32() { return go::Uint32::from_raw((uint32_t)next(state())); }
inline void Seed(go::Int64 seed) { state() = (uint64_t)seed.v; }
}  // namespace go_rand
