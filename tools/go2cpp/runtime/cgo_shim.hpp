// cgo as the translated shim files of go/ see it (oracle/_ref_gpu) — TEST INFRASTRUCTURE, a RUNTIME SHIM of tools/go2cpp.
//
// `import "C"` in go/gpu_batch.go, gpu_transform.go and gpu_entropy.go means include/knz_gpu.h. Here every C type the shim names is a
// Go-typed view of the real C type (same size and layout, checked below), every C function it calls forwards to the real entry point of
// libknz_gpu.so, and `unsafe` / `runtime.Pinner` / `runtime.SetFinalizer` are what they are to a program whose memory does not move.
#pragma once
#include "../../../include/knz_gpu.h"
#include "go_rt.hpp"

namespace go_unsafe {
struct Pointer {
    void* p = nullptr;
    Pointer() = default;
    Pointer(go::nil_t) {}
    explicit Pointer(void* x) : p(x) {}
    template <class T> explicit Pointer(T* x) : p((void*)x) {}
    friend bool operator==(const Pointer& a, go::nil_t) { return a.p == nullptr; }
    friend bool operator!=(const Pointer& a, go::nil_t) { return a.p != nullptr; }
    friend bool operator==(const Pointer& a, const Pointer& b) { return a.p == b.p; }
    friend bool operator!=(const Pointer& a, const Pointer& b) { return a.p != b.p; }
};
template <class T> inline go::Uintptr Sizeof(const T&) { return go::Uintptr::from_raw(sizeof(T)); }
}  // namespace go_unsafe

namespace go {
// T(unsafe.Pointer) and unsafe.Pointer(T): pointer conversions
template <class To> struct conv_ptr {
    static To from(const go_unsafe::Pointer& x) { return (To)x.p; }
};
template <class To, class = std::enable_if_t<std::is_pointer_v<To>>> inline To conv_from_unsafe(const go_unsafe::Pointer& x) { return (To)x.p; }
// a pointer to an array sliced: (*[N]T)(p)[:n:n]
template <class T, size_t N, class L, class H, class M> inline Slice<T> slice3(Array<T, N>* a, L lo, H hi, M mx) {
    int64_t l = bound(lo, 0), h = bound(hi, (int64_t)N), m = bound(mx, (int64_t)N);
    if (l < 0 || h < l || m < h || m > (int64_t)N) oob_slice(l, h, (int64_t)N);
    return Slice<T>(a->a + l, h - l, m - l);
}
template <class T, size_t N, class L, class H> inline Slice<T> slice(Array<T, N>* a, L lo, H hi) { return slice3(a, lo, hi, none); }
}  // namespace go

namespace go_C {
struct c_tag {};
using uint8_t = go::I<::uint8_t, c_tag>;
using uint16_t = go::I<::uint16_t, c_tag>;
using uint32_t = go::I<::uint32_t, c_tag>;
using uint64_t = go::I<::uint64_t, c_tag>;
using int8_t = go::I<::int8_t, c_tag>;
using int32_t = go::I<::int32_t, c_tag>;
using int64_t = go::I<::int64_t, c_tag>;
using int_ = go::I<int, c_tag>;
using uint_ = go::I<unsigned, c_tag>;
using size_t = go::I<::size_t, c_tag>;
using char_ = go::I<char, c_tag>;

struct knz_cfg {
    knz_cfg* operator->() { return this; }
    uint64_t transform; uint32_t entropy; uint32_t block_size; uint32_t checksum_bits; uint32_t bs_version; int32_t device; uint32_t flags;
};
static_assert(sizeof(knz_cfg) == sizeof(::knz_cfg) && offsetof(knz_cfg, flags) == offsetof(::knz_cfg, flags), "go_C::knz_cfg mirrors knz_cfg");
struct knz_block {
    knz_block* operator->() { return this; }
    uint8_t* src; uint32_t src_len; uint8_t* dst; uint32_t dst_cap; uint64_t out_bits; uint32_t post_len; uint8_t skip_flags; uint8_t mode; uint16_t reserved;
    uint64_t checksum; int32_t status; int32_t reserved2;
};
static_assert(sizeof(knz_block) == sizeof(::knz_block) && offsetof(knz_block, status) == offsetof(::knz_block, status) &&
              offsetof(knz_block, out_bits) == offsetof(::knz_block, out_bits), "go_C::knz_block mirrors knz_block");

constexpr go::U KNZ_FLAG_SKIP_BLOCKS = go::U(::KNZ_FLAG_SKIP_BLOCKS);
constexpr go::U KNZ_SKIP = go::U(::KNZ_SKIP);

inline go::String GoString(const char* s) { return go::String(s ? s : ""); }
inline go_unsafe::Pointer calloc(size_t n, size_t sz) { return go_unsafe::Pointer(go::alloc_zero(n.v * sz.v)); }   // (given back with the call's arena)
inline void free(go_unsafe::Pointer) {}

inline int_ knz_open(knz_cfg* cfg, go_unsafe::Pointer* handle) { return int_::from_raw(::knz_open((const ::knz_cfg*)cfg, &handle->p)); }
inline int_ knz_open_devices(knz_cfg* cfg, int32_t* ordinals, int_ n, go_unsafe::Pointer* handle) {
    return int_::from_raw(::knz_open_devices((const ::knz_cfg*)cfg, (const ::int32_t*)ordinals, n.v, &handle->p));
}
inline int_ knz_device_count() { return int_::from_raw(::knz_device_count()); }
inline int_ knz_close(go_unsafe::Pointer h) { return int_::from_raw(::knz_close(h.p)); }
inline const char* knz_last_error(go_unsafe::Pointer h) { return ::knz_last_error(h.p); }
inline int_ knz_encode_blocks(go_unsafe::Pointer h, knz_block* blocks, int_ n) { return int_::from_raw(::knz_encode_blocks(h.p, (::knz_block*)blocks, n.v)); }
inline int_ knz_decode_blocks(go_unsafe::Pointer h, knz_block* blocks, int_ n) { return int_::from_raw(::knz_decode_blocks(h.p, (::knz_block*)blocks, n.v)); }
inline int_ knz_supports(uint64_t t, uint32_t e) { return int_::from_raw(::knz_supports(t.v, e.v)); }
inline uint32_t knz_max_encoded_len(uint64_t t, uint32_t n) { return uint32_t::from_raw(::knz_max_encoded_len(t.v, n.v)); }
inline int_ knz_transform_forward(go_unsafe::Pointer h, uint64_t t, uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint32_t* out_n) {
    return int_::from_raw(::knz_transform_forward(h.p, t.v, (const ::uint8_t*)src, n.v, (::uint8_t*)dst, cap.v, (::uint32_t*)out_n));
}
inline int_ knz_transform_inverse(go_unsafe::Pointer h, uint64_t t, uint8_t* src, uint32_t n, uint8_t* dst, uint32_t cap, uint32_t* out_n) {
    return int_::from_raw(::knz_transform_inverse(h.p, t.v, (const ::uint8_t*)src, n.v, (::uint8_t*)dst, cap.v, (::uint32_t*)out_n));
}
inline int_ knz_entropy_encode(go_unsafe::Pointer h, uint32_t type, uint8_t* src, uint32_t n, uint8_t* bits, uint64_t cap_bytes, uint64_t* out_bits) {
    return int_::from_raw(::knz_entropy_encode(h.p, type.v, (const ::uint8_t*)src, n.v, (::uint8_t*)bits, cap_bytes.v, (::uint64_t*)out_bits));
}
inline int_ knz_entropy_decode(go_unsafe::Pointer h, uint32_t type, uint8_t* bits, uint64_t n_bytes, uint8_t* dst, uint32_t n, uint64_t* used_bits) {
    return int_::from_raw(::knz_entropy_decode(h.p, type.v, (const ::uint8_t*)bits, n_bytes.v, (::uint8_t*)dst, n.v, (::uint64_t*)used_bits));
}
}  // namespace go_C
