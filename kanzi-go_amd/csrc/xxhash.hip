// Block checksums of kanzi bitstream v6 (-x / --checksum=32|64) on gfx950: hash.XXHash32.Hash / hash.XXHash64.Hash with seed
// 0x4B414E5A ('KANZ'), v2/hash/XXHash32.go:51-108, XXHash64.go:51-120, called by encodingTask.encode before the transforms
// (v2/io/CompressedStream.go:760-767) and by decodingTask.decode on the decoded block (:1992-2007, mismatch = ERR_CRC_CHECK).
// The four accumulators of a stripe are independent, each is a serial chain v = rotl(v + x * P2, r) * P1 over the whole block:
// one wave per block, lane & 3 = accumulator (the other lanes mirror, no divergence), tiles staged through LDS with coalesced
// 16-byte loads, the LDS reads of 8 steps issued ahead of the dependent arithmetic. XXHash64.go merges its accumulators with
// 32-bit style shift pairs on 64-bit words (NOT standard XXH64); restated literally.
#include "bits.h"

#define KNZ_XXH_SEED 0x4B414E5Au
#define KNZ_XXH_TILE 8192

struct XxhArgs {
    uint32_t nblocks;
    const uint64_t* ptr;          // [nblocks] absolute address of the block bytes
    const uint32_t* len;          // [nblocks]
    uint64_t* cksum;              // [nblocks] verify == 0: result; verify == 1: expected value (from the block header)
    int32_t* status;              // [nblocks] verify == 1: set to ERR_CRC_CHECK on mismatch (blocks already failed are skipped)
    const uint8_t* mode;          // verify == 1: block mode bytes (nullptr = none)
    uint32_t bits;                // 32 or 64
    uint32_t verify;
};

__device__ __forceinline__ uint32_t knz_rotl32(uint32_t v, int r) { return (v << r) | (v >> (32 - r)); }
__device__ __forceinline__ uint64_t knz_rotl64(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }

// stages cnt bytes (multiple of 16 unless last) of the block into LDS
__device__ __forceinline__ void knz_xxh_stage(uint8_t* s_tile, const uint8_t* src, uint32_t cnt, int lane) {
    if ((((uintptr_t)src) & 15) == 0) {
        for (uint32_t i = lane; i < (cnt >> 4); i += 64) ((uint4*)s_tile)[i] = ((const uint4*)src)[i];
        for (uint32_t i = (cnt & ~15u) + lane; i < cnt; i += 64) s_tile[i] = src[i];
    } else {
        for (uint32_t i = lane; i < cnt; i += 64) s_tile[i] = src[i];
    }
}

__global__ __launch_bounds__(64) void knz_xxhash_kernel(XxhArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[KNZ_XXH_TILE];
    const int lane = threadIdx.x, c = lane & 3;
    const uint32_t b = blockIdx.x;
    if (a.verify && a.status[b] != 0) return;
    const uint8_t* src = (const uint8_t*)a.ptr[b];
    const uint32_t n = a.len[b];
    uint64_t result;
    if (a.bits == 32) {
        const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
        const uint32_t seed = KNZ_XXH_SEED;
        uint32_t v = c == 0 ? seed + P1 + P2 : (c == 1 ? seed + P2 : (c == 2 ? seed : seed - P1));
        const uint32_t stripes = n >= 16 ? ((n - 16) >> 4) + 1 : 0;          // while (n <= end16) (:68-80)
        for (uint32_t s0 = 0; s0 < stripes; s0 += KNZ_XXH_TILE / 16) {
            const uint32_t cnt = min((uint32_t)(KNZ_XXH_TILE / 16), stripes - s0);
            wave_sync();
            knz_xxh_stage(s_tile, src + (size_t)s0 * 16, cnt * 16, lane);
            wave_sync();
            const uint32_t* w = (const uint32_t*)s_tile + c;
            uint32_t s = 0;
            for (; s + 8 <= cnt; s += 8) {
                uint32_t x[8];
#pragma unroll
                for (int u = 0; u < 8; u++) x[u] = w[4 * (s + u)];
#pragma unroll
                for (int u = 0; u < 8; u++) v = knz_rotl32(v + x[u] * P2, 13) * P1;
            }
            for (; s < cnt; s++) v = knz_rotl32(v + w[4 * s] * P2, 13) * P1;
        }
        const uint32_t v1 = wave_shfl(v, 0), v2 = wave_shfl(v, 1), v3 = wave_shfl(v, 2), v4 = wave_shfl(v, 3);
        uint32_t h = stripes ? knz_rotl32(v1, 1) + knz_rotl32(v2, 7) + knz_rotl32(v3, 12) + knz_rotl32(v4, 18) : seed + P5;
        h += n;
        uint32_t p = stripes * 16;
        for (; p + 4 <= n; p += 4) {
            const uint32_t x = (uint32_t)src[p] | ((uint32_t)src[p + 1] << 8) | ((uint32_t)src[p + 2] << 16) | ((uint32_t)src[p + 3] << 24);
            h = knz_rotl32(h + x * P3, 17) * P4;
        }
        for (; p < n; p++) h = knz_rotl32(h + (uint32_t)src[p] * P5, 11) * P1;
        h ^= h >> 15; h *= P2; h ^= h >> 13; h *= P3;
        result = h ^ (h >> 16);
    } else {
        const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                       P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
        const uint64_t seed = KNZ_XXH_SEED;
        uint64_t v = c == 0 ? seed + P1 + P2 : (c == 1 ? seed + P2 : (c == 2 ? seed : seed - P1));
        const uint32_t stripes = n >= 32 ? ((n - 32) >> 5) + 1 : 0;
        for (uint32_t s0 = 0; s0 < stripes; s0 += KNZ_XXH_TILE / 32) {
            const uint32_t cnt = min((uint32_t)(KNZ_XXH_TILE / 32), stripes - s0);
            wave_sync();
            knz_xxh_stage(s_tile, src + (size_t)s0 * 32, cnt * 32, lane);
            wave_sync();
            const uint64_t* w = (const uint64_t*)s_tile + c;
            uint32_t s = 0;
            for (; s + 4 <= cnt; s += 4) {
                uint64_t x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) x[u] = w[4 * (s + u)];
#pragma unroll
                for (int u = 0; u < 4; u++) v = knz_rotl64(v + x[u] * P2, 31) * P1;
            }
            for (; s < cnt; s++) v = knz_rotl64(v + w[4 * s] * P2, 31) * P1;
        }
        const uint64_t v1 = wave_shfl64(v, 0), v2 = wave_shfl64(v, 1), v3 = wave_shfl64(v, 2), v4 = wave_shfl64(v, 3);
        uint64_t h;
        if (stripes) {
            h = ((v1 << 1) | (v1 >> 31)) + ((v2 << 7) | (v2 >> 25)) + ((v3 << 12) | (v3 >> 20)) + ((v4 << 18) | (v4 >> 14));
            const uint64_t vs[4] = {v1, v2, v3, v4};
#pragma unroll
            for (int u = 0; u < 4; u++) { h ^= knz_rotl64(vs[u] * P2, 31) * P1; h = h * P1 + P4; }
        } else h = seed + P5;
        h += n;
        uint32_t p = stripes * 32;
        for (; p + 8 <= n; p += 8) {
            uint64_t x = 0;
            for (int q = 0; q < 8; q++) x |= (uint64_t)src[p + q] << (8 * q);
            h ^= knz_rotl64(x * P2, 31) * P1;
            h = knz_rotl64(h, 27) * P1 + P4;
        }
        for (; p + 4 <= n; p += 4) {
            const uint64_t x = (uint64_t)src[p] | ((uint64_t)src[p + 1] << 8) | ((uint64_t)src[p + 2] << 16) | ((uint64_t)src[p + 3] << 24);
            h ^= x * P1;
            h = knz_rotl64(h, 23) * P2 + P3;
        }
        for (; p < n; p++) { h += (uint64_t)src[p] * P5; h = knz_rotl64(h, 11) * P1; }
        h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3;
        result = h ^ (h >> 32);
    }
    if (lane == 0) {
        if (a.verify) { if (a.cksum[b] != result) a.status[b] = KNZ_ERR_CRC_CHECK; }
        else a.cksum[b] = result;
    }
}
