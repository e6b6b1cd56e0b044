// Device-wide sort / scan / select primitives used by the suffix sort and the inverse BWT: the hand-written kernels of prims.hip
// (round 1 used rocPRIM's; the last A/B is profiles/r02_config4_bench_{own,rocprim}_primitives.json: forward stage 88.0 vs 100.7 ms). Under the
// execution-model emulator (tests/emu, CPU container, test infrastructure only) the entry points are served by plain loops over host
// memory unless KNZ_EMU_PRIMS=kernels asks for the real kernels (one dedicated test: the loops keep the BWT cases of the CPU suite fast).
#pragma once
#include "knz_internal.h"
#include "prims.hip"

#ifndef KNZ_HIP_EMU
// (kin, vin) are scratch: clobbered. The result is in (kout, vout).
static inline int knz_sort_pairs_u64(DevBuf& tmp, uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout, size_t n,
                                     unsigned b0, unsigned b1, hipStream_t st) {
    return knz_own_sort_pairs<uint64_t>(tmp, kin, kout, vin, vout, n, b0, b1, st);
}
static inline int knz_sort_pairs_u32(DevBuf& tmp, uint32_t* kin, uint32_t* kout, uint32_t* vin, uint32_t* vout, size_t n,
                                     unsigned b0, unsigned b1, hipStream_t st) {
    return knz_own_sort_pairs<uint32_t>(tmp, kin, kout, vin, vout, n, b0, b1, st);
}
static inline int knz_scan_max_u32(DevBuf& tmp, uint32_t* in, uint32_t* out, size_t n, hipStream_t st) { return knz_own_scan_max_u32(tmp, in, out, n, st); }
// out_idx[k] = k-th index i in [0,n) with flags[i] != 0 ; *d_count = number selected
static inline int knz_select_flagged(DevBuf& tmp, const uint8_t* flags, uint32_t* out_idx, uint32_t* d_count, size_t n, hipStream_t st) {
    return knz_own_select_flagged(tmp, flags, out_idx, d_count, n, st);
}
#else
#include <algorithm>
#include <numeric>
#include <vector>
template <typename K>
static inline int knz_sort_pairs_emu(K* kin, K* kout, uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1) {
    std::vector<size_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    const K mask = (b1 - b0) >= sizeof(K) * 8 ? ~(K)0 : ((((K)1) << (b1 - b0)) - 1);
    std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return ((kin[x] >> b0) & mask) < ((kin[y] >> b0) & mask); });
    for (size_t i = 0; i < n; i++) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
    return 0;
}
static inline bool knz_emu_kernels() { return getenv("KNZ_EMU_PRIMS") != nullptr; }
static inline int knz_sort_pairs_u64(DevBuf& tmp, uint64_t* kin, uint64_t* kout, uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1, hipStream_t st) {
    if (knz_emu_kernels()) return knz_own_sort_pairs<uint64_t>(tmp, kin, kout, vin, vout, n, b0, b1, st);
    return knz_sort_pairs_emu(kin, kout, vin, vout, n, b0, b1);
}
static inline int knz_sort_pairs_u32(DevBuf& tmp, uint32_t* kin, uint32_t* kout, uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1, hipStream_t st) {
    if (knz_emu_kernels()) return knz_own_sort_pairs<uint32_t>(tmp, kin, kout, vin, vout, n, b0, b1, st);
    return knz_sort_pairs_emu(kin, kout, vin, vout, n, b0, b1);
}
static inline int knz_scan_max_u32(DevBuf& tmp, uint32_t* in, uint32_t* out, size_t n, hipStream_t st) {
    if (knz_emu_kernels()) return knz_own_scan_max_u32(tmp, in, out, n, st);
    uint32_t m = 0;
    for (size_t i = 0; i < n; i++) { m = in[i] > m ? in[i] : m; out[i] = m; }
    return 0;
}
static inline int knz_select_flagged(DevBuf& tmp, const uint8_t* flags, uint32_t* out_idx, uint32_t* d_count, size_t n, hipStream_t st) {
    if (knz_emu_kernels()) return knz_own_select_flagged(tmp, flags, out_idx, d_count, n, st);
    uint32_t c = 0;
    for (size_t i = 0; i < n; i++) if (flags[i]) out_idx[c++] = (uint32_t)i;
    *d_count = c;
    return 0;
}
#endif
