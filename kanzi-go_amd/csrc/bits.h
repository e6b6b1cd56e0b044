// Bit-string helpers shared by the gfx950 kernels.
// kanzi bit streams are MSB-first (v2/bitstream/DefaultOutputBitStream.go:78-98): bit i of a stream is
// bit (7 - i%8) of byte i/8. Kernels work on 32-bit "BE words": word w = bytes 4w..4w+3 read big-endian,
// so stream bit 32w+k is bit (31-k) of the word; memory stores are byte-swapped 32-bit stores.
#pragma once
#include "wave.h"

#define KNZ_HUF_CHUNK 16384          // _HUF_MAX_CHUNK_SIZE, HuffmanCodec.go:30
#define KNZ_HUF_MAXLEN 12            // _HUF_MAX_SYMBOL_SIZE_V4, HuffmanCodec.go:31
#define KNZ_UNITS_PER_CHUNK 5        // [header+varints][frag0][frag1][frag2][frag3+tail]
#define KNZ_U0_BYTES 512             // scratch bytes reserved for unit 0
#define KNZ_FRAG_BYTES 6160          // 4096 symbols * 12 bits = 6144 bytes, + tail 3 + pad
#define KNZ_CHUNK_STRIDE (KNZ_U0_BYTES + 4 * KNZ_FRAG_BYTES)
// rANS order 0 (16 KiB chunks): u0 header | u1 varint(size)+4 states | u2 renormalisation words (right aligned) + tail
#define KNZ_ANS_CHUNK 16384           // _DEFAULT_ANS0_CHUNK_SIZE, ANSRangeCodec.go:33
#define KNZ_ANS_U1_OFF 512
#define KNZ_ANS_PAY_OFF 576
#define KNZ_ANS_PAY_CAP 24704         // >= 16384 symbols * 12 bits / 8 + 16 (a symbol costs at most log2(4096) bits)
#define KNZ_ANS_SLOT (KNZ_ANS_PAY_OFF + KNZ_ANS_PAY_CAP + 64)

__device__ __forceinline__ uint32_t knz_bswap32(uint32_t v) { return __builtin_bswap32(v); }

// BE word i of a byte buffer that is 4-byte aligned
__device__ __forceinline__ uint32_t knz_load_be32(const uint8_t* base, int64_t word) {
    return knz_bswap32(((const uint32_t*)base)[word]);
}

// 32 stream bits starting at bit position `bit` (bit may be negative or run past nbits: missing bits read 0)
// of a 4-byte aligned MSB-first buffer holding nbits valid bits.
__device__ __forceinline__ uint32_t knz_fetch32(const uint8_t* base, int64_t bit, int64_t nbits) {
    int64_t q = bit >> 5;          // floor
    int r = (int)(bit & 31);
    int64_t nwords = (nbits + 31) >> 5;
    uint32_t w0 = (q >= 0 && q < nwords) ? knz_load_be32(base, q) : 0u;
    uint32_t w1 = (q + 1 >= 0 && q + 1 < nwords) ? knz_load_be32(base, q + 1) : 0u;
    uint32_t v = r ? ((w0 << r) | (w1 >> (32 - r))) : w0;
    // clear bits at or beyond nbits
    int64_t valid = nbits - bit;   // number of leading bits of v that are inside the string (if bit >= 0)
    if (valid <= 0) return 0u;
    if (valid < 32) v &= ~(0xFFFFFFFFu >> valid);
    return v;
}

// Same for a unit that starts at an arbitrary BYTE offset of a 4-byte aligned slot: `rel` is relative to the unit
// start (may be negative), the unit holds nbits valid bits.
__device__ __forceinline__ uint32_t knz_fetch32_unit(const uint8_t* slot, uint32_t unitByteOff, int64_t rel, int64_t nbits) {
    if (rel >= nbits || rel <= -32) return 0u;
    const int64_t abs = (int64_t)unitByteOff * 8 + rel;     // absolute bit inside the slot, may be < 0 only if rel < 0
    const int64_t q = abs >> 5;
    const int r = (int)(abs & 31);
    const uint32_t w0 = q >= 0 ? knz_load_be32(slot, q) : 0u;
    const uint32_t w1 = (q + 1) >= 0 ? knz_load_be32(slot, q + 1) : 0u;
    uint32_t v = r ? ((w0 << r) | (w1 >> (32 - r))) : w0;
    if (rel < 0) v &= 0xFFFFFFFFu >> (-rel);                // bits before the unit start
    const int64_t valid = nbits - rel;                      // bits of v (from its MSB) that lie inside the unit
    if (valid < 32) v &= ~(0xFFFFFFFFu >> valid);
    return v;
}

// Serial MSB-first bit writer into a zeroed BE-word buffer (LDS or global). One thread only.
struct KnzBitWriter {
    uint32_t* words;
    uint32_t pos;      // bits written
    __device__ __forceinline__ void init(uint32_t* w) { words = w; pos = 0; }
    __device__ __forceinline__ void put(uint32_t value, uint32_t count) { // count in [0..32]
        if (count == 0) return;
        if (count < 32) value &= (1u << count) - 1u;
        uint32_t w = pos >> 5, off = pos & 31;
        uint32_t room = 32 - off;
        if (count <= room) {
            words[w] |= value << (room - count);
        } else {
            uint32_t rem = count - room;
            words[w] |= value >> rem;
            words[w + 1] |= value << (32 - rem);
        }
        pos += count;
    }
};

// EntropyUtils.go:264-275 WriteVarInt through a KnzBitWriter
__device__ __forceinline__ void knz_put_varint(KnzBitWriter& bw, uint32_t value) {
    while (value >= 128) { bw.put(0x80 | (value & 0x7F), 8); value >>= 7; }
    bw.put(value, 8);
}

// Signed Exp-Golomb emit word (len<<9 | bits): closed form of ExpGolombCodec.go:45-62
// (v as int8, n=|v|+1, L=floor(log2 n): (n<<1|sign) on 2L+2 bits; v == 0 is the single bit '1').
__device__ __forceinline__ uint32_t knz_expg_signed(int v) {
    if (v == 0) return (1u << 9) | 1u;
    uint32_t sign = v < 0 ? 1u : 0u;
    uint32_t n = (uint32_t)(v < 0 ? -v : v) + 1u;
    uint32_t L = 31u - (uint32_t)__builtin_clz(n);
    return ((2u * L + 2u) << 9) | ((n << 1) | sign);
}

// Order-0 histogram of n bytes into 4 per-wave private LDS histograms (256 threads). Global.go:226-251.
// Coalesced 16 B per lane when src is 16-byte aligned. Caller zeroes hist and synchronises before/after.
__device__ __forceinline__ void knz_histogram_256t(const uint8_t* src, uint32_t n, uint32_t (*hist)[256], int tid) {
    uint32_t* h = hist[tid >> 6];
    const uint32_t nvec = n >> 4;
    if ((((uintptr_t)src) & 15) == 0) {
        const uint4* v = (const uint4*)src;
        for (uint32_t i = tid; i < nvec; i += 256) {
            uint4 x = v[i];
            uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                atomicAdd(&h[w[j] & 255], 1u); atomicAdd(&h[(w[j] >> 8) & 255], 1u);
                atomicAdd(&h[(w[j] >> 16) & 255], 1u); atomicAdd(&h[w[j] >> 24], 1u);
            }
        }
        for (uint32_t i = (nvec << 4) + tid; i < n; i += 256) atomicAdd(&h[src[i]], 1u);
    } else {
        for (uint32_t i = tid; i < n; i += 256) atomicAdd(&h[src[i]], 1u);
    }
}

// One wave, 4 histograms (picked by lane & 3; hist is [4][256], zeroed by the caller): the bytes of one fragment of a Huffman chunk
__device__ __forceinline__ void knz_histogram_64t_x4(const uint8_t* src, uint32_t n, uint32_t (*hist)[256], int lane) {
    uint32_t* h = hist[lane & 3];
    if ((((uintptr_t)src) & 15) == 0) {
        const uint32_t nvec = n >> 4;
        const uint4* v = (const uint4*)src;
        for (uint32_t i = lane; i < nvec; i += 64) {
            uint4 x = v[i];
            uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                atomicAdd(&h[w[j] & 255], 1u); atomicAdd(&h[(w[j] >> 8) & 255], 1u);
                atomicAdd(&h[(w[j] >> 16) & 255], 1u); atomicAdd(&h[w[j] >> 24], 1u);
            }
        }
        for (uint32_t i = (nvec << 4) + lane; i < n; i += 64) atomicAdd(&h[src[i]], 1u);
    } else {
        for (uint32_t i = lane; i < n; i += 64) atomicAdd(&h[src[i]], 1u);
    }
}

// Same with 16 histograms (4 per wave, picked by lane & 3): frequent symbols (spaces, zeros) make the lanes of a wave collide
// on one LDS counter, 4 counters per wave cut that serialisation by 4. hist is [16][256], zeroed by the caller.
__device__ __forceinline__ void knz_histogram_256t_x16(const uint8_t* src, uint32_t n, uint32_t (*hist)[256], int tid) {
    uint32_t* h = hist[((tid >> 6) << 2) | (tid & 3)];
    const uint32_t nvec = n >> 4;
    if ((((uintptr_t)src) & 15) == 0) {
        const uint4* v = (const uint4*)src;
        for (uint32_t i = tid; i < nvec; i += 256) {
            uint4 x = v[i];
            uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                atomicAdd(&h[w[j] & 255], 1u); atomicAdd(&h[(w[j] >> 8) & 255], 1u);
                atomicAdd(&h[(w[j] >> 16) & 255], 1u); atomicAdd(&h[w[j] >> 24], 1u);
            }
        }
        for (uint32_t i = (nvec << 4) + tid; i < n; i += 256) atomicAdd(&h[src[i]], 1u);
    } else {
        for (uint32_t i = tid; i < n; i += 256) atomicAdd(&h[src[i]], 1u);
    }
}

// per-lane unaligned 8-byte / 4-byte loads (one global_load_dwordx2 / dword: gfx950 runs with unaligned access enabled)
struct __attribute__((packed)) KnzPacked64 { uint64_t v; };
__device__ __forceinline__ uint64_t knz_vle64(const uint8_t* p) { return ((const KnzPacked64*)p)->v; }
struct __attribute__((packed)) KnzPacked128 { uint32_t x, y, z, w; };
struct __attribute__((packed)) KnzPacked32 { uint32_t v; };
__device__ __forceinline__ uint32_t knz_vle32(const uint8_t* p) { return ((const KnzPacked32*)p)->v; }
