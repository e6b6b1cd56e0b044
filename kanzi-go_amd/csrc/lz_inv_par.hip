// LZ / LZX inverse without the per-block chain (LZXCodec.inverseV6, v2/transform/LZCodec.go:621-778; readLengthLZ :214-231).
//
// The reference walks the tokens of a block one after the other with four cursors (literals, tokens, distances, match-length
// extensions) and copies as it goes; lz.hip does the same with one wave per block (~0.8 us per token whatever the GPU does
// meanwhile). Nothing in the format forces that order except three things, and each has a data-parallel form:
//   * the cursors. Token bytes are independent of each other. The distance cursor advances by the token's `f` bits alone: a
//     prefix sum. The match-length extensions are a stream of self-delimiting records (1, 3 or 4 bytes by their first byte):
//     record boundaries are found by composing, per 16-byte stretch, the map "bytes still owed by the previous record ->
//     (bytes owed to the next stretch, records started)" over a 4-state machine. Literal lengths >= 7 are the one cursor
//     that depends on the DATA at the cursor (the extension sits inline in the literal stream): a sparse chain over the tokens
//     that carry one (one scalar load and ~15 scalar instructions each), everything between two of them is a prefix sum;
//   * the two repeat distances. (repd0, repd1) after a token is a function of (repd0, repd1) before it: an explicit distance
//     x gives (x, d0), a repeat of repd0 gives (d0, d0), a repeat of repd1 gives (d1, d0). Functions whose two outputs are each
//     "a constant, the incoming d0 or the incoming d1" are closed under composition, so every token's distance comes out of one
//     prefix scan with that operator;
//   * the copies (a match may read bytes that an earlier match of the same block wrote). Every output byte is either a
//     literal (its position in the literal stream is known from the prefix sums) or the copy of an EARLIER output byte: a
//     per-byte source map (4 bytes per output byte) is resolved by following the map (workgroups are dispatched in ascending
//     order, so most sources are already final when a byte is visited; what is not follows at most 8 hops per pass and writes
//     back where it got to, which shortens every later path through it: two such passes), then one gather pass reads the literal
//     bytes (and follows whatever is still open to its end: every path ends at a literal).
// A block takes this path only if it is well formed (every cursor ends where the header says, every distance passes the
// reference's sanity checks :732-735, the literal cursor reaches the end of the literals at the last token and not before);
// any other block - damaged streams - is left to the one-wave kernel of lz.hip, which keeps the reference's behaviour token by
// token (counter KNZ_COUNTER_LZ_INV_SERIAL_BLOCKS).
#pragma once
#include "bits.h"

#define KNZ_LZI_SEG 2048u                     // tokens per segment: 256 threads x 8
#define KNZ_LZI_LIT 0x80000000u               // source map: literal byte, low bits = position in the block's input
#define KNZ_LZI_SEL0 0xFFFFFFFEu              // repeat-distance maps: "the incoming repd0" / "the incoming repd1"
#define KNZ_LZI_SEL1 0xFFFFFFFFu
// The source map is shortened by pointer doubling before the gather follows what is left: passes x hops per pass. Measured on S-silesia (51 x 4 MiB,
// jump + gather ms): 2 x 8 (rounds 3-5) 4.84 + 0.43, 2 x 4 3.76 + 0.81, 3 x 2 3.19 + 0.81, 4 x 1 2.45 + 1.47, 5 x 1 2.94 + 0.83, none 0 + 22.8
// (`tools/gpu/r06_jump.sh`). Any choice is exact: the gather follows every path to its literal.
#ifndef KNZ_LZI_HOPS
#define KNZ_LZI_HOPS 1
#endif
#ifndef KNZ_LZI_JUMP_PASSES
#define KNZ_LZI_JUMP_PASSES 5
#endif

// geo[16 b + ..]
enum { LZI_TK0 = 0, LZI_NTOK = 1, LZI_M0 = 2, LZI_ML0 = 3, LZI_COUNT = 4, LZI_MINMATCH = 5, LZI_MAXDIST = 6, LZI_PAR = 7,
       LZI_NLEXT = 8, LZI_NMEXT = 9, LZI_DBYTES = 10, LZI_NREC = 11, LZI_OUT = 12 };

struct LziArgs {
    LzArgs a;
    uint32_t segs;                 // token segments per block covered by the grid
    uint32_t* geo;                 // [nblocks][16]
    const uint32_t* tok_base;      // [nblocks] first slot of the block in the per-token arrays (nTok + 2 slots per block)
    uint32_t* t_a;                 // pass A: short literal bytes in front of the token (an extended length counts 7) ; pass B: input position of its literals
    uint32_t* t_b;                 // pass A: distance bytes in front of the token ; pass B: its distance
    uint32_t* t_c;                 // pass A: literal-length extensions in front of the token ; pass B: its literal length
    uint32_t* t_d;                 // pass A: match-length extensions in front of the token ; pass B: output position of its literals ([nTok] = total)
    uint32_t* seg;                 // [nblocks][segs][8]
    uint32_t* lx_g;                // per literal-length extension: 13 + t_a of its token + its ordinal in the block (the "+ 1" of every extension in front of it, taken off the chain)
    uint32_t* lx_c;                // extension bytes + extension values in front of it ([n] = total): size and value of one = the difference
    uint32_t* ml_val;              // values of the match-length extension records, in order
    uint32_t* map; uint64_t map_stride;
    uint32_t* unfinished;
    uint8_t* serial;               // [nblocks] 1 = the block goes to knz_lz_inverse_kernel
};

__device__ __forceinline__ uint64_t knz_lzi_load8(const uint8_t* p, uint32_t avail) {
    if (avail >= 8) return knz_vle64(p);
    uint64_t v = 0;
    for (uint32_t j = 0; j < avail; j++) v |= (uint64_t)p[j] << (8 * j);
    return v;
}

// what a token byte says on its own (:653-719). The last token of a block carries literals only.
struct LziTok { uint32_t lit, lext, dbytes, mext, mbase, rep; };
__device__ __forceinline__ LziTok knz_lzi_token(uint32_t t, bool last) {
    LziTok k;
    k.lit = t >> 5; k.lext = k.lit == 7 ? 1u : 0u;
    const uint32_t f = t & 0x18u;
    k.dbytes = last ? 0u : (f >> 3);
    k.mbase = f == 0 ? (t & 3u) : (t & 7u);
    k.mext = last ? 0u : (f == 0 ? (k.mbase == 3u) : (k.mbase == 7u));
    k.rep = last ? 3u : (f != 0 ? 0u : ((t & 4u) ? 2u : 1u));          // 0 explicit, 1 repd0, 2 repd1, 3 no match
    return k;
}

// one thread per block: the header (:635-662) and whether the block can take this path at all
__global__ __launch_bounds__(64) void knz_lzi_header_kernel(LziArgs g) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.a.nblocks) return;
    uint32_t* G = g.geo + 16 * (size_t)b;
    for (int i = 0; i < 16; i++) G[i] = 0;
    g.serial[b] = 0;
    if (!g.a.active[b]) return;
    g.serial[b] = 1;                                                        // until the checks of this path have all passed
    const uint64_t count = g.a.in_len[b];
    const uint8_t* src = (const uint8_t*)g.a.in_ptr[b];
    if (count < 13) return;
    const uint64_t tk0 = knz_le32(src), m0 = tk0 + knz_le32(src + 4), ml0 = m0 + knz_le32(src + 8);
    if (tk0 < 13 || tk0 > count || m0 > count || ml0 > count || m0 == tk0 || count >= 0x7FFFFFF0ull) return;
    G[LZI_TK0] = (uint32_t)tk0; G[LZI_NTOK] = (uint32_t)(m0 - tk0); G[LZI_M0] = (uint32_t)m0; G[LZI_ML0] = (uint32_t)ml0;
    G[LZI_COUNT] = (uint32_t)count;
    G[LZI_MINMATCH] = ((src[12] >> 1) & 7u) + 2u;
    G[LZI_MAXDIST] = (src[12] & 1u) ? KNZ_LZ_MAX_DIST2 : KNZ_LZ_MAX_DIST1;
    G[LZI_PAR] = 1;
}

__device__ __forceinline__ uint32_t knz_lzi_wg_sum(uint32_t v, uint32_t* s_w) {      // 256 threads; result valid in every thread
    v = wave_reduce_add(v);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t r = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return r;
}
__device__ __forceinline__ uint32_t knz_lzi_wg_scan_excl(uint32_t v, uint32_t* s_w) {  // 256 threads
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_scan_incl(v);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < 4; k++) base += k < w ? s_w[k] : 0u;
    __syncthreads();
    return base + incl - v;
}

// ---- pass A: what the token bytes alone give ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knz_lzi_a_count_kernel(LziArgs g) {
    __shared__ uint32_t s_w[4];
    const uint32_t b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const uint32_t* G = g.geo + 16 * (size_t)b;
    const uint32_t nTok = G[LZI_NTOK];
    if (!G[LZI_PAR] || s * KNZ_LZI_SEG >= nTok) return;
    const uint32_t k0 = s * KNZ_LZI_SEG + tid * 8;
    uint32_t sl = 0, db = 0, lx = 0, mx = 0;
    if (k0 < nTok) {
        const uint8_t* src = (const uint8_t*)g.a.in_ptr[b];
        const uint64_t w = knz_lzi_load8(src + G[LZI_TK0] + k0, nTok - k0);
        for (uint32_t j = 0; j < 8 && k0 + j < nTok; j++) {
            const LziTok k = knz_lzi_token((uint32_t)(w >> (8 * j)) & 0xFFu, k0 + j == nTok - 1);
            sl += k.lit; db += k.dbytes; lx += k.lext; mx += k.mext;
        }
    }
    const uint32_t p0 = knz_lzi_wg_sum(sl | (db << 16), s_w), p1 = knz_lzi_wg_sum(lx | (mx << 16), s_w);   // <= 14336, 6144, 2048, 2048 per segment
    if (tid == 0) {
        uint32_t* S = g.seg + ((size_t)b * g.segs + s) * 8;
        S[0] = p0 & 0xFFFFu; S[1] = p0 >> 16; S[2] = p1 & 0xFFFFu; S[3] = p1 >> 16;
    }
}

// one wave per block: exclusive sums over the segments, totals into the geometry
__global__ __launch_bounds__(64) void knz_lzi_a_offsets_kernel(LziArgs g) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    uint32_t* G = g.geo + 16 * (size_t)b;
    if (!G[LZI_PAR]) return;
    const uint32_t ns = (G[LZI_NTOK] + KNZ_LZI_SEG - 1) / KNZ_LZI_SEG;
    uint32_t c[4] = {0, 0, 0, 0};
    for (uint32_t s0 = 0; s0 < ns; s0 += 64) {
        const uint32_t s = s0 + lane;
        uint32_t* S = g.seg + ((size_t)b * g.segs + s) * 8;
        uint32_t v[4] = {0, 0, 0, 0};
        if (s < ns) { v[0] = S[0]; v[1] = S[1]; v[2] = S[2]; v[3] = S[3]; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t incl = wave_scan_incl(v[q]);
            if (s < ns) S[q] = c[q] + incl - v[q];
            c[q] += wave_shfl(incl, 63);
        }
    }
    if (lane == 0) { G[LZI_NLEXT] = c[2]; G[LZI_NMEXT] = c[3]; G[LZI_DBYTES] = c[1]; }
}

__global__ __launch_bounds__(256) void knz_lzi_a_apply_kernel(LziArgs g) {
    __shared__ uint32_t s_w[4];
    const uint32_t b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    const uint32_t* G = g.geo + 16 * (size_t)b;
    const uint32_t nTok = G[LZI_NTOK];
    if (!G[LZI_PAR] || s * KNZ_LZI_SEG >= nTok) return;
    const uint32_t k0 = s * KNZ_LZI_SEG + tid * 8;
    uint32_t sl = 0, db = 0, lx = 0, mx = 0;
    uint64_t w = 0;
    if (k0 < nTok) {
        const uint8_t* src = (const uint8_t*)g.a.in_ptr[b];
        w = knz_lzi_load8(src + G[LZI_TK0] + k0, nTok - k0);
        for (uint32_t j = 0; j < 8 && k0 + j < nTok; j++) {
            const LziTok k = knz_lzi_token((uint32_t)(w >> (8 * j)) & 0xFFu, k0 + j == nTok - 1);
            sl += k.lit; db += k.dbytes; lx += k.lext; mx += k.mext;
        }
    }
    const uint32_t e0 = knz_lzi_wg_scan_excl(sl | (db << 16), s_w), e1 = knz_lzi_wg_scan_excl(lx | (mx << 16), s_w);
    if (k0 >= nTok) return;
    const uint32_t* S = g.seg + ((size_t)b * g.segs + s) * 8;
    uint32_t a = S[0] + (e0 & 0xFFFFu), d = S[1] + (e0 >> 16), c = S[2] + (e1 & 0xFFFFu), m = S[3] + (e1 >> 16);
    const size_t tb = g.tok_base[b];
    for (uint32_t j = 0; j < 8 && k0 + j < nTok; j++) {
        const LziTok k = knz_lzi_token((uint32_t)(w >> (8 * j)) & 0xFFu, k0 + j == nTok - 1);
        const size_t i = tb + k0 + j;
        g.t_a[i] = a; g.t_b[i] = d; g.t_c[i] = c; g.t_d[i] = m;
        if (k.lext) g.lx_g[tb + c] = 13u + a + c;
        a += k.lit; d += k.dbytes; c += k.lext; m += k.mext;
    }
}

// ---- literal lengths >= 7: the extension sits in the literal stream at the literal cursor (:657-661), so its position depends on
// every extension in front of it: one wave per block walks them. The chain is  C -> byte at (13 + g[i] + C) -> C + 1 + byte  (the 1 per extension is
// folded into g[i] by the kernel that writes it: the walk carries C minus the number of extensions it has passed, one add less per step) and it is
// kept on the scalar unit: the g[i] arrive through the scalar cache eight at a time; the literal region is held as two 256-byte
// windows in two vector registers (lane L = dword L; the window behind the current one is loaded while the current one is walked),
// so that the byte at the cursor is one v_readlane and three scalar instructions away, with no memory access on the chain at all.
// The chain stores C in front of every extension (64 steps per coalesced store); size and value of an extension are recovered from
// the differences, which is unambiguous for the records an encoder writes (emitLengthLZ :193-212: 1 byte below 254, 3 bytes below
// 65790, else 4): anything else leaves the block to the one-wave kernel.
__device__ __forceinline__ void knz_lzi_ext_from_diff(uint32_t d, uint32_t& sz, uint32_t& val) {
    sz = d < 255u ? 1u : (d < 65793u ? 3u : 4u);
    val = d - sz;
}
__global__ __launch_bounds__(64) void knz_lzi_litext_chain_kernel(LziArgs g) {
    __shared__ uint32_t s_hist[16 * 64];
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    uint32_t* G = g.geo + 16 * (size_t)b;
    if (!G[LZI_PAR]) return;
    const uint32_t E = wave_uniform(G[LZI_NLEXT]), tk0 = wave_uniform(G[LZI_TK0]), count = wave_uniform(G[LZI_COUNT]);
    const uint8_t* src = (const uint8_t*)g.a.in_ptr[b];
    const size_t tb = g.tok_base[b];
    const uint8_t* lg = (const uint8_t*)(g.lx_g + tb);
    uint32_t* lc = g.lx_c + tb;
    uint32_t C = 0, hist = 0;
    bool bad = ((uintptr_t)src & 3) != 0;                                    // (the pipeline's regions are 16-byte aligned)
    // windows: w0 = the 64 dwords at wlo, w1 = the 64 dwords behind them (zero behind the end of the block)
    auto load_win = [&](uint32_t at) -> uint32_t { const uint32_t o = at + 4 * lane; return o < count ? *(const uint32_t*)(src + o) : 0u; };
    uint32_t wlo = 12;                                                        // the literals start at 13
    uint32_t w0 = bad ? 0u : load_win(wlo), w1 = bad ? 0u : load_win(wlo + 256);
    // one step. What a damaged stream can do to the walk (a cursor behind the literals, a sum that wraps) is not tested per step: the
    // window loads are bounded by the block, the walk takes exactly E steps, and pass B checks every token's literal range against the
    // end of the literals (a block that fails there goes to the one-wave kernel).
    auto window = [&](uint32_t pos) -> uint32_t {                              // the byte at pos, through the windows
        uint32_t o = pos - wlo;
        if (__builtin_expect(o >= 256, 0)) {                                   // (pos only grows, except when a group of steps is walked again: then it lies in front and both windows are loaded)
            if (o < 512) { w0 = w1; wlo += 256; }
            else { wlo = pos & ~3u; w0 = load_win(wlo); }                      // a jump over the whole next window
            w1 = load_win(wlo + 256);
            o = pos - wlo;
        }
        const uint32_t dw = wave_readlane(w0, o >> 2);
        return (dw >> (8 * (o & 3))) & 0xFFu;
    };
    auto advance = [&](uint32_t gj) {
        const uint32_t pos = gj + C;                                           // (g[] holds 13 + the short literal bytes in front + the extensions in front; C their bytes and values minus one each)
        const uint32_t b0 = window(pos);
        uint32_t d = b0;                                                       // (bytes + value of this extension, minus one)
        if (__builtin_expect(b0 >= 254, 0)) {                                  // rare: the three / four byte forms, read where they stand
            uint32_t b1 = 0, b2 = 0, b3 = 0;
            if (pos + 4 <= count) { b1 = wave_uniform(src[pos + 1]); b2 = wave_uniform(src[pos + 2]); b3 = wave_uniform(src[pos + 3]); } else bad = true;
            if (b0 == 254) d = 2u + 254u + (b1 << 8) + b2;
            else { const uint32_t y = (b1 << 16) + (b2 << 8) + b3; if (y < 65535u) bad = true; d = 3u + 255u + y; }
        }
        C += d;
    };
    // the same step taken for a one-byte extension without looking: the largest byte of sixteen steps is looked at once, and if one of them was a three or
    // four byte form (a literal run of 261 bytes or more) the sixteen are walked again with the step above. A compare and a branch less on the chain per step.
    uint32_t big = 0;
    auto advance1 = [&](uint32_t gj) { const uint32_t b0 = window(gj + C); big = max(big, b0); C += b0; };
    auto step = [&](uint32_t gj, uint32_t j) { hist = lane == j ? C : hist; advance(gj); };
    // (the C in front of step L goes to lane L of `hist` with one v_writelane_b32; lane == j ? C : hist is a move, a compare and a select per step, a quarter of the walk's instructions)
#define KNZ_LZI_STEP(G, L) { hist = wave_writelane_at<(L)>(hist, C); advance(G); }
#define KNZ_LZI_STEP1(G, L) { hist = wave_writelane_at<(L)>(hist, C); advance1(G); }
#define KNZ_LZI_16(S, Q) \
        S(a.x, 16 * (Q)) S(a.y, 16 * (Q) + 1) S(a.z, 16 * (Q) + 2) S(a.w, 16 * (Q) + 3) \
        S(b2.x, 16 * (Q) + 4) S(b2.y, 16 * (Q) + 5) S(b2.z, 16 * (Q) + 6) S(b2.w, 16 * (Q) + 7) \
        S(c2.x, 16 * (Q) + 8) S(c2.y, 16 * (Q) + 9) S(c2.z, 16 * (Q) + 10) S(c2.w, 16 * (Q) + 11) \
        S(d2.x, 16 * (Q) + 12) S(d2.y, 16 * (Q) + 13) S(d2.z, 16 * (Q) + 14) S(d2.w, 16 * (Q) + 15)
#define KNZ_LZI_STEPS16(Q) if (n >= 16u * (Q) + 16u) { \
        const uint8_t* gp = lg + 4 * (size_t)(i0 + 16u * (Q)); \
        const knz_u32x4 a = wave_sload_u32x4(gp), b2 = wave_sload_u32x4(gp + 16), c2 = wave_sload_u32x4(gp + 32), d2 = wave_sload_u32x4(gp + 48); \
        const uint32_t C0 = C; big = 0; \
        KNZ_LZI_16(KNZ_LZI_STEP1, Q) \
        if (__builtin_expect(big >= 254u, 0)) { C = C0; KNZ_LZI_16(KNZ_LZI_STEP, Q) } }
    for (uint32_t i0 = 0; i0 < E && !bad; i0 += 64) {
        const uint32_t n = min(64u, E - i0);
        // the g[i] through the scalar cache, sixteen at a time: a scalar load is waited for where it is issued (the compiler drains lgkmcnt in front of
        // every v_readlane of the walk), so its ~150 ns sit on the chain once per load - once per sixteen steps now, once per four until round 6
        KNZ_LZI_STEPS16(0) KNZ_LZI_STEPS16(1) KNZ_LZI_STEPS16(2) KNZ_LZI_STEPS16(3)
        uint32_t j = n & ~15u;
        for (; j + 4 <= n; j += 4) {
            const knz_u32x4 a = wave_sload_u32x4(lg + 4 * (size_t)(i0 + j));
            step(a.x, j); step(a.y, j + 1); step(a.z, j + 2); step(a.w, j + 3);
        }
        for (; j < n; j++) step(wave_sload_u32(lg + 4 * (size_t)(i0 + j)), j);
        // The 64 cursors of the group wait in LDS and leave for memory sixteen groups at a time: the compiler drains vmcnt in front of every
        // v_readlane of the walk (the windows are vector loads), so a store to memory is waited for by the very next step - once per 1024 steps this way
        const uint32_t slot = (i0 >> 6) & 15u;
        s_hist[slot * 64 + lane] = hist + i0 + lane;                           // (the ones come back here)
        if (slot == 15u || i0 + 64 >= E) {
            wave_sync_lds();
            const uint32_t base = i0 - slot * 64;
            for (uint32_t k = 0; k <= slot; k++) { const uint32_t i = base + k * 64 + lane; if (i < E) lc[i] = s_hist[k * 64 + lane]; }
        }
    }
    if (C >= 0x40000000u) bad = true;
    if (lane == 0) { if (bad) G[LZI_PAR] = 0; else lc[E] = C + E; }
}
#undef KNZ_LZI_STEPS16
#undef KNZ_LZI_16
#undef KNZ_LZI_STEP1
#undef KNZ_LZI_STEP

// ---- match-length extensions: a stream of records of 1, 3 or 4 bytes (readLengthLZ :214-231) ------------------------------------
struct LziRecMap { uint32_t exit; uint64_t cnt; };                      // per entry state s = 0..3: exit state (2 bits each), records started (16 bits each)
__device__ __forceinline__ LziRecMap knz_lzi_rec_compose(const LziRecMap& A, const LziRecMap& B) {   // A first
    LziRecMap r; r.exit = 0; r.cnt = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const uint32_t e = (A.exit >> (2 * s)) & 3u;
        r.exit |= ((B.exit >> (2 * e)) & 3u) << (2 * s);
        r.cnt |= (((A.cnt >> (16 * s)) + (B.cnt >> (16 * e))) & 0xFFFFull) << (16 * s);
    }
    return r;
}
__global__ __launch_bounds__(256) void knz_lzi_mlen_parse_kernel(LziArgs g) {
    __shared__ uint32_t s_exit[4];
    __shared__ uint64_t s_cnt[4];
    __shared__ uint32_t s_carry[2];
    const uint32_t b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint32_t* G = g.geo + 16 * (size_t)b;
    if (!G[LZI_PAR]) return;
    const uint32_t M = G[LZI_NMEXT];
    if (M == 0) return;
    const uint8_t* reg = (const uint8_t*)g.a.in_ptr[b] + G[LZI_ML0];
    const uint32_t R = G[LZI_COUNT] - G[LZI_ML0];
    uint32_t* out = g.ml_val + g.tok_base[b];
    uint32_t st = 0, nrec = 0;                                          // bytes the previous tile's last record still owns, records so far
    bool bad = false;
    for (uint32_t t0 = 0; t0 < R && nrec < M; t0 += 4096) {
        const uint32_t off = t0 + tid * 16;
        uint8_t by[16];
#pragma unroll
        for (int i = 0; i < 16; i++) by[i] = off + i < R ? reg[off + i] : 0;
        LziRecMap m; m.exit = 0; m.cnt = 0;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            uint32_t pos = s, c = 0;
            while (pos < 16) { const uint32_t v = by[pos]; c += off + pos < R ? 1u : 0u; pos += v < 254 ? 1u : (v == 254 ? 3u : 4u); }
            m.exit |= (pos - 16) << (2 * s);
            m.cnt |= (uint64_t)c << (16 * s);
        }
        // inclusive scan of the maps over the wave, then the four waves in order
        LziRecMap inc = m;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            LziRecMap o; o.exit = wave_shfl(inc.exit, (int)((lane - d) & 63)); o.cnt = wave_shfl64(inc.cnt, (int)((lane - d) & 63));
            if ((int)lane >= d) inc = knz_lzi_rec_compose(o, inc);
        }
        if (lane == 63) { s_exit[wv] = inc.exit; s_cnt[wv] = inc.cnt; }
        __syncthreads();
        uint32_t ws = st, wn = nrec;                                       // state / records at the start of this wave
        for (uint32_t k = 0; k < wv; k++) { wn += (uint32_t)(s_cnt[k] >> (16 * ws)) & 0xFFFFu; ws = (s_exit[k] >> (2 * ws)) & 3u; }
        LziRecMap prev; prev.exit = wave_shfl(inc.exit, (int)((lane - 1) & 63)); prev.cnt = wave_shfl64(inc.cnt, (int)((lane - 1) & 63));
        uint32_t es = ws, en = wn;
        if (lane > 0) { en = wn + ((uint32_t)(prev.cnt >> (16 * ws)) & 0xFFFFu); es = (prev.exit >> (2 * ws)) & 3u; }
        // replay the stretch from its true entry state
        for (uint32_t pos = es; pos < 16 && off + pos < R; ) {
            const uint32_t v = by[pos];
            const uint32_t sz = v < 254 ? 1u : (v == 254 ? 3u : 4u);
            if (en < M) {
                if (off + pos + sz > R) bad = true;
                else {
                    const uint8_t* p = reg + off + pos;
                    out[en] = v < 254 ? v : (v == 254 ? 254u + ((uint32_t)p[1] << 8) + p[2] : 255u + ((uint32_t)p[1] << 16) + ((uint32_t)p[2] << 8) + p[3]);
                }
            }
            en++; pos += sz;
        }
        if (tid == 255) { s_carry[0] = (inc.exit >> (2 * ws)) & 3u; s_carry[1] = wn + ((uint32_t)(inc.cnt >> (16 * ws)) & 0xFFFFu); }
        __syncthreads();
        st = s_carry[0]; nrec = s_carry[1];
        __syncthreads();
    }
    if (bad) G[LZI_PAR] = 0;
    if (tid == 0) { G[LZI_NREC] = nrec; if (nrec < M) G[LZI_PAR] = 0; }
}

// ---- pass B: lengths, positions, distances -----------------------------------------------------------------------------------
struct LziRep { uint32_t c0, c1; };                                       // (repd0, repd1) behind a stretch of tokens as a function of the pair in front of it
__device__ __forceinline__ uint32_t knz_lzi_rep_res(uint32_t v, const LziRep& A) { return v == KNZ_LZI_SEL0 ? A.c0 : (v == KNZ_LZI_SEL1 ? A.c1 : v); }
__device__ __forceinline__ LziRep knz_lzi_rep_compose(const LziRep& A, const LziRep& B) {   // A first
    LziRep r; r.c0 = knz_lzi_rep_res(B.c0, A); r.c1 = knz_lzi_rep_res(B.c1, A); return r;
}
__device__ __forceinline__ LziRep knz_lzi_rep_token(uint32_t rep, uint32_t dist) {        // :721-722 after the choice of `dist`
    LziRep r;
    if (rep == 0) { r.c0 = dist; r.c1 = KNZ_LZI_SEL0; }
    else if (rep == 1) { r.c0 = KNZ_LZI_SEL0; r.c1 = KNZ_LZI_SEL0; }
    else if (rep == 2) { r.c0 = KNZ_LZI_SEL1; r.c1 = KNZ_LZI_SEL0; }
    else { r.c0 = KNZ_LZI_SEL0; r.c1 = KNZ_LZI_SEL1; }
    return r;
}
// inclusive scan of one map per thread over the 256 threads of a workgroup (ordered: the operator does not commute)
__device__ __forceinline__ LziRep knz_lzi_wg_rep_scan(LziRep v, LziRep* s_r, LziRep& total) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        LziRep o; o.c0 = wave_shfl(v.c0, (int)((lane - d) & 63)); o.c1 = wave_shfl(v.c1, (int)((lane - d) & 63));
        if ((int)lane >= d) v = knz_lzi_rep_compose(o, v);
    }
    if (lane == 63) s_r[w] = v;
    __syncthreads();
    LziRep acc; acc.c0 = KNZ_LZI_SEL0; acc.c1 = KNZ_LZI_SEL1;
    LziRep tot = acc;
    for (uint32_t k = 0; k < 4; k++) { if (k < w) acc = knz_lzi_rep_compose(acc, s_r[k]); tot = knz_lzi_rep_compose(tot, s_r[k]); }
    __syncthreads();
    total = tot;
    return knz_lzi_rep_compose(acc, v);
}

// the 8 tokens of a thread: literal length, match length (0 behind the last token), repeat kind, explicit distance
struct LziTok8 { uint32_t lit[8], mlen[8], dist[8]; uint8_t rep[8]; uint32_t n; bool bad; };
__device__ __forceinline__ void knz_lzi_b_tokens(const LziArgs& g, uint32_t b, const uint32_t* G, uint32_t k0, LziTok8& T) {
    const uint32_t nTok = G[LZI_NTOK];
    const uint8_t* src = (const uint8_t*)g.a.in_ptr[b];
    const size_t tb = g.tok_base[b];
    T.n = k0 < nTok ? min(8u, nTok - k0) : 0u;
    T.bad = false;
    if (T.n == 0) return;
    const uint64_t w = knz_lzi_load8(src + G[LZI_TK0] + k0, nTok - k0);
    const uint32_t minMatch = G[LZI_MINMATCH];
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) if (j < T.n) {
        const LziTok k = knz_lzi_token((uint32_t)(w >> (8 * j)) & 0xFFu, k0 + j == nTok - 1);
        const size_t i = tb + k0 + j;
        if (k.lext) { const uint32_t x = g.t_c[i]; uint32_t sz, val; knz_lzi_ext_from_diff(g.lx_c[tb + x + 1] - g.lx_c[tb + x], sz, val); T.lit[j] = 7u + val; }
        else T.lit[j] = k.lit;
        T.rep[j] = (uint8_t)k.rep;
        T.mlen[j] = k.rep == 3 ? 0u : k.mbase + minMatch + (k.mext ? g.ml_val[tb + g.t_d[i]] : 0u);
        uint32_t d = 0;
        if (k.rep == 0) {
            const uint32_t off = G[LZI_M0] + g.t_b[i];
            if (off + k.dbytes > G[LZI_ML0]) T.bad = true;
            else { const uint8_t* p = src + off; d = p[0]; if (k.dbytes >= 2) d = (d << 8) | p[1]; if (k.dbytes == 3) d = (d << 8) | p[2]; }
        }
        T.dist[j] = d;
    }
}

__global__ __launch_bounds__(256) void knz_lzi_b_count_kernel(LziArgs g) {
    __shared__ uint32_t s_w[4];
    __shared__ LziRep s_r[4];
    const uint32_t b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    uint32_t* G = g.geo + 16 * (size_t)b;
    // (other workgroups of the block clear the flag while this kernel runs: the workgroup reads it ONCE, so that all its waves reach the barriers below or none)
    __shared__ uint32_t s_par;
    if (tid == 0) s_par = G[LZI_PAR];
    __syncthreads();
    if (!s_par || s * KNZ_LZI_SEG >= G[LZI_NTOK]) return;
    LziTok8 T;
    knz_lzi_b_tokens(g, b, G, s * KNZ_LZI_SEG + tid * 8, T);
    uint32_t len = 0;
    LziRep m; m.c0 = KNZ_LZI_SEL0; m.c1 = KNZ_LZI_SEL1;
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) if (j < T.n) {
        if (T.lit[j] > 0x3FFFFFFFu || T.mlen[j] > 0x3FFFFFFFu || len > 0x3FFFFFFFu) T.bad = true;
        len += T.lit[j] + T.mlen[j];
        m = knz_lzi_rep_compose(m, knz_lzi_rep_token(T.rep[j], T.dist[j]));
    }
    // per-thread sums stay below 2^31 (checked above), the segment sum is accumulated in 64 bits through two halves
    const uint32_t lo = knz_lzi_wg_sum(len & 0xFFFFu, s_w), hi = knz_lzi_wg_sum(len >> 16, s_w);
    LziRep tot;
    (void)knz_lzi_wg_rep_scan(m, s_r, tot);
    if (T.bad) G[LZI_PAR] = 0;
    if (tid == 0) {
        const uint64_t sum = ((uint64_t)hi << 16) + lo;
        uint32_t* S = g.seg + ((size_t)b * g.segs + s) * 8;
        S[4] = sum > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)sum; S[5] = tot.c0; S[6] = tot.c1;
        if (sum > 0x7FFFFFF0ull) G[LZI_PAR] = 0;
    }
}

// one wave per block: output position and (repd0, repd1) in front of every segment
__global__ __launch_bounds__(64) void knz_lzi_b_offsets_kernel(LziArgs g) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    uint32_t* G = g.geo + 16 * (size_t)b;
    if (!G[LZI_PAR]) return;
    const uint32_t ns = (G[LZI_NTOK] + KNZ_LZI_SEG - 1) / KNZ_LZI_SEG;
    uint64_t pos = 0;
    LziRep st; st.c0 = G[LZI_COUNT]; st.c1 = G[LZI_COUNT];                   // repd0 = repd1 = count (:650-651)
    for (uint32_t s0 = 0; s0 < ns; s0 += 64) {
        const uint32_t s = s0 + lane;
        uint32_t* S = g.seg + ((size_t)b * g.segs + s) * 8;
        uint32_t v = 0;
        LziRep m; m.c0 = KNZ_LZI_SEL0; m.c1 = KNZ_LZI_SEL1;
        if (s < ns) { v = S[4]; m.c0 = S[5]; m.c1 = S[6]; }
        // sums in two 16-bit halves: the 32-bit wave scan cannot overflow
        const uint32_t ilo = wave_scan_incl(v & 0xFFFFu), ihi = wave_scan_incl(v >> 16);
        const uint64_t incl = ((uint64_t)ihi << 16) + ilo;
        LziRep inc = m;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            LziRep o; o.c0 = wave_shfl(inc.c0, (int)((lane - d) & 63)); o.c1 = wave_shfl(inc.c1, (int)((lane - d) & 63));
            if ((int)lane >= d) inc = knz_lzi_rep_compose(o, inc);
        }
        LziRep prev; prev.c0 = wave_shfl(inc.c0, (int)((lane - 1) & 63)); prev.c1 = wave_shfl(inc.c1, (int)((lane - 1) & 63));
        if (lane == 0) { prev.c0 = KNZ_LZI_SEL0; prev.c1 = KNZ_LZI_SEL1; }
        const LziRep entry = knz_lzi_rep_compose(st, prev);
        const uint64_t ex = pos + incl - v;
        if (s < ns) { S[4] = ex > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)ex; S[5] = entry.c0; S[6] = entry.c1; }
        LziRep last; last.c0 = wave_shfl(inc.c0, 63); last.c1 = wave_shfl(inc.c1, 63);
        st = knz_lzi_rep_compose(st, last);
        pos += ((uint64_t)wave_shfl(ihi, 63) << 16) + wave_shfl(ilo, 63);
    }
    if (lane == 0) {
        if (pos > (uint64_t)g.a.out_cap || pos > 0x7FFFFFF0ull) G[LZI_PAR] = 0;
        G[LZI_OUT] = pos > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)pos;
    }
}

__global__ __launch_bounds__(256) void knz_lzi_b_apply_kernel(LziArgs g) {
    __shared__ uint32_t s_w[4];
    __shared__ LziRep s_r[4];
    const uint32_t b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    uint32_t* G = g.geo + 16 * (size_t)b;
    const uint32_t nTok = G[LZI_NTOK];
    __shared__ uint32_t s_par;                                            // (read once per workgroup, see knz_lzi_b_count_kernel)
    if (tid == 0) s_par = G[LZI_PAR];
    __syncthreads();
    if (!s_par || s * KNZ_LZI_SEG >= nTok) return;
    const uint32_t k0 = s * KNZ_LZI_SEG + tid * 8;
    LziTok8 T;
    knz_lzi_b_tokens(g, b, G, k0, T);
    uint32_t len = 0;
    LziRep m; m.c0 = KNZ_LZI_SEL0; m.c1 = KNZ_LZI_SEL1;
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) if (j < T.n) { len += T.lit[j] + T.mlen[j]; m = knz_lzi_rep_compose(m, knz_lzi_rep_token(T.rep[j], T.dist[j])); }
    const uint32_t* S = g.seg + ((size_t)b * g.segs + s) * 8;
    uint32_t dpos = S[4] + knz_lzi_wg_scan_excl(len, s_w);                   // (the block total fits 31 bits: checked by the offsets kernel)
    LziRep tot;
    const LziRep incl = knz_lzi_wg_rep_scan(m, s_r, tot);
    // the pair in front of this thread's tokens = (segment entry) o (threads in front of it): shift the inclusive scan by one thread
    LziRep before; before.c0 = wave_shfl(incl.c0, (int)((tid - 1) & 63)); before.c1 = wave_shfl(incl.c1, (int)((tid - 1) & 63));
    __shared__ LziRep s_last[4];
    if ((tid & 63) == 63) s_last[tid >> 6] = incl;
    __syncthreads();
    if ((tid & 63) == 0) { if (tid == 0) { before.c0 = KNZ_LZI_SEL0; before.c1 = KNZ_LZI_SEL1; } else before = s_last[(tid >> 6) - 1]; }
    LziRep st; st.c0 = S[5]; st.c1 = S[6];
    st = knz_lzi_rep_compose(st, before);                                    // constants: the segment entry holds no selector
    bool bad = T.bad;
    const size_t tb = g.tok_base[b];
    const uint32_t tk0 = G[LZI_TK0], srcEnd = tk0 - 13u, maxDist = G[LZI_MAXDIST];
    const int64_t dstEnd = (int64_t)g.a.out_cap - 16;
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) if (j < T.n) {
        const size_t i = tb + k0 + j;
        const uint32_t k = k0 + j;
        const bool last = k == nTok - 1;
        // input position of the literals: 13 + short literal bytes + everything the extensions in front added + this token's own extension bytes
        const uint32_t lxi = g.t_c[i];
        uint32_t spos = 13u + g.t_a[i] + g.lx_c[tb + lxi];
        // (is this token one that carries an extension? its literal field says so: recompute from the token byte)
        const uint32_t tokByte = ((const uint8_t*)g.a.in_ptr[b])[tk0 + k];
        if ((tokByte >> 5) == 7u) { uint32_t sz, val; knz_lzi_ext_from_diff(g.lx_c[tb + lxi + 1] - g.lx_c[tb + lxi], sz, val); spos += sz; }
        const uint32_t lit = T.lit[j];
        // the reference stops at the first token whose literals reach srcEnd (:672-674) and wants the cursor at the end of the literals then (:771)
        if ((uint64_t)spos + lit > tk0) bad = true;
        if (tokByte >= 32) {
            const bool stop = spos + lit >= srcEnd;
            if (stop != last) bad = true;
            if (last && spos + lit != tk0) bad = true;
        } else if (last) bad = true;
        uint32_t dist = 0;
        if (!last) {
            const LziRep after = knz_lzi_rep_compose(st, knz_lzi_rep_token(T.rep[j], T.dist[j]));
            dist = after.c0;
            st = after;
            const int64_t mstart = (int64_t)dpos + lit, mEnd = mstart + T.mlen[j];
            if (dist == 0 || dist > maxDist || mstart - (int64_t)dist < 0 || mEnd > dstEnd) bad = true;   // :732-735 (dist 0: lz.hip fails the block)
        }
        g.t_a[i] = spos; g.t_b[i] = dist; g.t_c[i] = lit; g.t_d[i] = dpos;
        dpos += lit + T.mlen[j];
        if (last) g.t_d[i + 1] = dpos;
    }
    if (bad) G[LZI_PAR] = 0;
}

// which blocks go where; results of the blocks that took this path
__global__ __launch_bounds__(64) void knz_lzi_finalize_kernel(LziArgs g) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.a.nblocks) return;
    const uint32_t* G = g.geo + 16 * (size_t)b;
    if (!g.a.active[b]) { g.serial[b] = 0; return; }
    if (G[LZI_PAR]) { g.serial[b] = 0; g.a.out_len[b] = G[LZI_OUT]; g.a.ok[b] = 1; }
    else g.serial[b] = 1;
}

// ---- the copies ----------------------------------------------------------------------------------------------------------------
// source map: one wave per 64 tokens writes, for every output byte of its tokens, where the byte comes from
__global__ __launch_bounds__(256) void knz_lzi_map_kernel(LziArgs g) {
    __shared__ uint32_t s_dst[4][65], s_src[4][64], s_lit[4][64], s_dist[4][64];
    const uint32_t b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t* G = g.geo + 16 * (size_t)b;
    const uint32_t nTok = G[LZI_NTOK];
    const uint32_t base = (blockIdx.x * 4 + wv) * 64;
    if (!G[LZI_PAR] || base >= nTok) return;
    const size_t tb = g.tok_base[b];
    const uint32_t nv = min(64u, nTok - base);
    if (lane < nv) {
        const size_t i = tb + base + lane;
        s_dst[wv][lane] = g.t_d[i]; s_src[wv][lane] = g.t_a[i]; s_lit[wv][lane] = g.t_c[i]; s_dist[wv][lane] = g.t_b[i];
    }
    if (lane == 0) s_dst[wv][nv] = g.t_d[tb + base + nv];
    wave_sync();
    uint32_t* map = g.map + (size_t)b * g.map_stride;
    const uint32_t D0 = s_dst[wv][0], D1 = s_dst[wv][nv];
    uint32_t t = 0;
    for (uint32_t p = D0 + lane; p < D1; p += 64) {
        while (p >= s_dst[wv][t + 1]) t++;
        uint32_t o = p - s_dst[wv][t];
        const uint32_t lit = s_lit[wv][t];
        uint32_t v;
        if (o < lit) v = KNZ_LZI_LIT | (s_src[wv][t] + o);
        else {
            o -= lit;
            const uint32_t d = s_dist[wv][t];
            // an overlapping match repeats its first `dist` bytes (:738-755: every byte is read after it was written)
            v = p - o - d + (o < d ? o : (d == 1 ? 0u : o % d));
        }
        map[p] = v;
    }
}

// follows the map of 4 output bytes per thread for at most KNZ_LZI_HOPS hops each and writes back how far it got
__global__ __launch_bounds__(256) void knz_lzi_jump_kernel(LziArgs g) {
    const uint32_t b = blockIdx.y;
    const uint32_t* G = g.geo + 16 * (size_t)b;
    if (!G[LZI_PAR]) return;
    const uint32_t total = G[LZI_OUT];
    const uint32_t p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    bool open = false;
    if (p0 < total) {
        uint32_t* map = g.map + (size_t)b * g.map_stride;
        const uint32_t n = min(4u, total - p0);
        uint32_t v[4];
        if (n == 4) { const uint4 q = *(const uint4*)(map + p0); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
        else for (uint32_t j = 0; j < 4; j++) v[j] = j < n ? map[p0 + j] : KNZ_LZI_LIT;
        bool changed = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t x = v[j];
            for (int hop = 0; hop < KNZ_LZI_HOPS && !(x & KNZ_LZI_LIT); hop++) x = map[x];
            changed |= x != v[j];
            v[j] = x;
            open |= !(x & KNZ_LZI_LIT);
        }
        if (changed) {
            if (n == 4) { uint4 q; q.x = v[0]; q.y = v[1]; q.z = v[2]; q.w = v[3]; *(uint4*)(map + p0) = q; }
            else for (uint32_t j = 0; j < n; j++) map[p0 + j] = v[j];
        }
    }
    (void)open;
}

__global__ __launch_bounds__(256) void knz_lzi_gather_kernel(LziArgs g) {
    const uint32_t b = blockIdx.y;
    const uint32_t* G = g.geo + 16 * (size_t)b;
    if (!G[LZI_PAR]) return;
    const uint32_t total = G[LZI_OUT];
    const uint32_t p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (p0 >= total) return;
    const uint32_t* map = g.map + (size_t)b * g.map_stride;
    const uint8_t* src = (const uint8_t*)g.a.in_ptr[b];
    uint8_t* dst = (uint8_t*)g.a.out_ptr[b];
    const uint32_t n = min(4u, total - p0);
    if (n == 4 && (((uintptr_t)dst) & 3) == 0) {
        const uint4 q = *(const uint4*)(map + p0);
        uint32_t v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; j++) while (!(v[j] & KNZ_LZI_LIT)) v[j] = map[v[j]];     // (whatever the jump passes left open: every path ends at a literal)
        const uint32_t w = (uint32_t)src[v[0] & 0x7FFFFFFFu] | ((uint32_t)src[v[1] & 0x7FFFFFFFu] << 8) | ((uint32_t)src[v[2] & 0x7FFFFFFFu] << 16) |
                           ((uint32_t)src[v[3] & 0x7FFFFFFFu] << 24);
        *(uint32_t*)(dst + p0) = w;
    } else {
        for (uint32_t j = 0; j < n; j++) { uint32_t x = map[p0 + j]; while (!(x & KNZ_LZI_LIT)) x = map[x]; dst[p0 + j] = src[x & 0x7FFFFFFFu]; }
    }
}
