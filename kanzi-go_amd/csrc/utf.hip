// UTF codec of kanzi bitstream v6 on gfx950: UTF-8 code points are replaced by 1- or 2-byte ranks (by decreasing frequency)
// behind a map of the code points. Replaces UTFCodec.Forward / Inverse / validateUTF / packUTF / unpackUTF1
// (v2/transform/UTFCodec.go:84-265, 268-383, 393-519, 521-546, 578-609).
//
// Unlike the list and dictionary transforms this one is data parallel once a block is known to be well-formed: a code point
// starts at every byte that is not a continuation byte. One workgroup of 16 waves per block (the batch has tens of blocks); passes 1 and 3
// stride the block with all of them, pass 2 runs on the first wave:
//   1. validation = the reference's statistics (validateUTF: no byte that never occurs in UTF-8, every lead byte followed by
//      a second byte of its legal range, at least 1/8 continuation bytes) plus the checks of its main loop (third / fourth
//      bytes, no stray continuation byte), all of them local to a position, and, in the same pass, the histogram of the
//      packed code points in a 2^22-entry table with the list of distinct code points built from the first touch (1- and 2-byte code
//      points are counted in LDS and join the table once per block);
//   2. ranks of the (at most 32767) code points by (frequency, code point) by counting, aliases written back into the table;
//   3. emission: per tile of 16 K positions the alias widths are prefix-summed (in the wave, then over the waves) and the bytes stored.
// The inverse finds the alias boundaries (a byte >= 128 takes the next byte with it, whatever that is) with a wave scan over
// the two-state automaton, prefix-sums the code point lengths and stores.
// A UTF stage behind another UTF stage that applied (ctx["dataType"] == DT_UTF8: the reference skips validateUTF then) takes
// the reference's sequential walk on lane 0 to mark the code point starts; that never happens in a useful sequence.
#include "bits.h"

#define KNZ_UTF_MIN_BLOCK 1024
#define KNZ_UTF_MAX_SYMS 32768
#define KNZ_UTF_MAP_LOG 22

struct UtfArgs {
    uint32_t nblocks;
    const uint64_t* in_ptr; const uint32_t* in_len;
    const uint64_t* out_ptr; uint32_t out_cap;
    uint32_t* out_len; int32_t* ok; const uint8_t* active;
    int32_t* alias_map;          // [nblocks << 22], zeroed by the host (forward)
    uint32_t* symlist;           // [nblocks * 32768] distinct code points in order of first touch (forward)
    uint32_t* ranks;             // [nblocks * 32768] sort position of symlist[j] (forward)
    uint64_t* inv_map;           // [nblocks * 32768] bytes of the code point | length << 32 (inverse)
    uint8_t* chain_bits;         // [nblocks * bits_stride] code point starts of the sequential walk (forward, dataType == UTF8 only)
    uint64_t bits_stride;
    uint8_t* blk_dt;             // [nblocks] ctx["dataType"] (internal/Global.go:26-40 numbering, text.hip): 0 undefined, 8 UTF8, ...; may be null
};

__device__ __forceinline__ uint32_t knz_utf_size(uint32_t b) {          // _UTF_SIZES :31-48
    return b < 0x80 ? 1u : (b < 0xC2 ? 0u : (b < 0xE0 ? 2u : (b < 0xF0 ? 3u : (b < 0xF5 ? 4u : 0u))));
}
// second byte legal behind lead byte a (validateUTF's pair rules :459-499; a is a lead byte of a 2..4 byte sequence)
__device__ __forceinline__ bool knz_utf_pair_ok(uint32_t a, uint32_t b) {
    if (a == 0xE0) return b >= 0xA0 && b <= 0xBF;
    if (a == 0xED) return b >= 0x80 && b <= 0x9F;
    if (a == 0xF0) return b >= 0x90 && b <= 0xBF;
    if (a == 0xF4) return b >= 0x80 && b <= 0x8F;
    return b >= 0x80 && b <= 0xBF;                                       // C2..DF, E1..EC, EE, EF, F1..F3
}
__device__ __forceinline__ uint32_t knz_utf_pack(uint32_t s, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {   // packUTF
    if (s == 1) return b0;
    if (s == 2) return (1u << 19) | (b0 << 8) | b1;
    if (s == 3) return (2u << 19) | ((b0 & 0x0F) << 12) | ((b1 & 0x3F) << 6) | (b2 & 0x3F);
    return (4u << 19) | ((b0 & 0x07) << 18) | ((b1 & 0x3F) << 12) | ((b2 & 0x3F) << 6) | (b3 & 0x3F);
}

#define KNZ_UTF_FWD_THREADS 1024
__global__ __launch_bounds__(KNZ_UTF_FWD_THREADS) void knz_utf_forward_kernel(UtfArgs a) {
    __shared__ uint32_t s_n;
    __shared__ uint32_t s_bad, s_cont;
    __shared__ int s_last;
    __shared__ uint64_t s_keys[1024];
    __shared__ uint32_t s_wt[KNZ_UTF_FWD_THREADS / 64];
    __shared__ int s_go, s_stop;
    __shared__ uint32_t s_c1[128], s_c2[2048];                           // counts of the 1- and 2-byte code points (the frequent ones) for the workgroup
    const int tid = threadIdx.x;                                         // passes 1 (validation + histogram) and 3 (emission) run on all 16 waves, pass 2 on wave 0
    const int lane = tid & 63;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const int count = (int)a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const uint32_t dt = a.blk_dt ? a.blk_dt[b] : 0u;
    if (count == 0) { if (tid == 0) { a.ok[b] = 1; a.out_len[b] = 0; } return; }
    // :93-114: block size, output size, ctx["dataType"]
    if (count < KNZ_UTF_MIN_BLOCK || (uint64_t)a.out_cap < (uint64_t)count + 8192 || (dt != KNZ_DT_UNDEFINED && dt != KNZ_DT_UTF8)) { if (tid == 0) { a.ok[b] = 0; a.out_len[b] = 0; } return; }
    int32_t* map = a.alias_map + ((size_t)b << KNZ_UTF_MAP_LOG);
    uint32_t* syms = a.symlist + (size_t)b * KNZ_UTF_MAX_SYMS;
    uint32_t* ranks = a.ranks + (size_t)b * KNZ_UTF_MAX_SYMS;
    uint8_t* cbits = a.chain_bits + (size_t)b * a.bits_stride;
    const bool chainMode = dt == KNZ_DT_UTF8;                                      // mustValidate == false
    int start = 0;
    if (src[1] == 0xEF && src[2] == 0xBB && src[3] == 0xBF) start = 3;   // BigEndian.Uint32(src) & 0x00FFFFFF == 0xEFBBBF (:118)
    else while (start < 4 && knz_utf_size(src[start]) == 0) start++;
    const int end = count - 4;                                           // code points start in [start, end)
    if (tid == 0) { s_n = 0; s_bad = 0; s_cont = 0; s_last = -1; s_go = 0; s_stop = 0; }
    for (int i = tid; i < 128 + 2048; i += KNZ_UTF_FWD_THREADS) { if (i < 128) s_c1[i] = 0; else s_c2[i - 128] = 0; }
    bool bad = false;
    if (chainMode && tid < 64) {
        // the reference's walk (:141-166), lane 0; marks the starts. (Only reachable with UTF twice in one sequence.)
        for (uint64_t i = lane; i < a.bits_stride; i += 64) cbits[i] = 0;
        wave_sync();
        __threadfence();
        int fail = 0;
        if (lane == 0) {
            for (int i = start; i < end;) {
                const uint32_t s = knz_utf_size(src[i]);
                if (s == 0 || (s == 3 && (src[i + 2] & 0xC0) != 0x80) || (s == 4 && ((src[i + 2] & 0xC0) != 0x80 || (src[i + 3] & 0xC0) != 0x80))) { fail = 1; break; }
                cbits[i >> 3] |= (uint8_t)(1u << (i & 7));
                i += (int)s;
            }
        }
        wave_sync();
        __threadfence();
        bad = wave_bcast((uint32_t)fail, 0) != 0;
        if (bad && lane == 0) s_bad = 1;
    }
    __syncthreads();
    if (s_bad) bad = true;
    // ---- pass 1: validation + histogram --------------------------------------------------------------------------------
    uint32_t cont = 0;                                                   // bytes in 80..BF inside [start, end) (sum2)
    int lastLead = -1;
    if (!bad) {
        for (int p0 = start; p0 < end; p0 += KNZ_UTF_FWD_THREADS) {
            if (wave_ballot(bad) != 0) break;                            // not UTF-8: the stage declines whatever follows (binary blocks stop within a row or two)
            const int p = p0 + tid;
            if (p < end) {
                const uint32_t b0 = src[p], b1 = src[p + 1], b2 = src[p + 2], b3 = src[p + 3];
                const uint32_t s = knz_utf_size(b0);
                bool isSym;
                if (chainMode) isSym = (cbits[p >> 3] >> (p & 7)) & 1;
                else {
                    isSym = s != 0;
                    if (b0 == 0xC0 || b0 == 0xC1 || b0 >= 0xF5) bad = true;                    // 1-byte rules
                    if (b0 >= 0x80 && b0 <= 0xBF) cont++;
                    if (s >= 2 && p + 1 < end && !knz_utf_pair_ok(b0, b1)) bad = true;           // 2-byte rules: pairs inside the region only
                    if (s >= 3 && (b2 & 0xC0) != 0x80) bad = true;                              // main loop (:146-149)
                    if (s == 4 && (b3 & 0xC0) != 0x80) bad = true;
                    if (s == 0) {
                        // a byte that starts nothing must belong to the sequence of a lead byte at most 3 positions in front of it
                        bool covered = false;
                        for (int d = 1; d <= 3 && p - d >= start; d++) if (knz_utf_size(src[p - d]) > (uint32_t)d) { covered = true; break; }
                        if (!covered) bad = true;
                    }
                }
                if (isSym) {
                    // (Round 4: code points of one and two bytes are counted in LDS and reach the table once per workgroup: a returned global
                    // atomic per code point - a thread waits for it before its next row, and the frequent ones share an address - was ~8 of the
                    // stage's 8.6 ms. Longer code points keep the table's own counters.)
                    if (s == 1) atomicAdd(&s_c1[b0], 1u);
                    else if (s == 2 && (b1 & 0xC0u) == 0x80u) atomicAdd(&s_c2[((b0 & 0x1Fu) << 6) | (b1 & 0x3Fu)], 1u);   // (a lead byte at the region's last position may be followed by anything: that pair keeps its exact bytes through the table's counter)
                    else {
                        const uint32_t val = knz_utf_pack(s, b0, b1, b2, b3);
                        if (atomicAdd(&map[val], 1) == 0) {
                            const uint32_t idx = atomicAdd(&s_n, 1u);
                            if (idx < KNZ_UTF_MAX_SYMS) syms[idx] = val;
                        }
                    }
                    lastLead = p;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 128 + 2048; i += KNZ_UTF_FWD_THREADS) {        // the LDS counts join the table (nobody else touches these entries)
        const uint32_t c = i < 128 ? s_c1[i] : s_c2[i - 128];
        if (c) {
            const uint32_t j = (uint32_t)i - 128u;
            const uint32_t val = i < 128 ? (uint32_t)i : knz_utf_pack(2, 0xC0u | (j >> 6), 0x80u | (j & 63u), 0, 0);
            map[val] = (int32_t)c;
            const uint32_t idx = atomicAdd(&s_n, 1u);
            if (idx < KNZ_UTF_MAX_SYMS) syms[idx] = val;
        }
    }
    {   // the 16 waves' findings
        const uint32_t cw = wave_reduce_add(cont);
        int lw = lastLead;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { const int o = (int)wave_shfl((uint32_t)lw, lane ^ d); lw = o > lw ? o : lw; }
        const uint64_t wb = wave_ballot(bad);
        if (lane == 0) { atomicAdd(&s_cont, cw); atomicMax(&s_last, lw); if (wb != 0) atomicOr(&s_bad, 1u); }
    }
    __threadfence();
    __syncthreads();
    const int maxTarget = count - count / 10;
    if (tid < 64) do {                                                   // wave 0: the decisions and the ranks; a decline leaves s_go at 0
    lastLead = s_last;
    const uint32_t n = s_n;
    const uint32_t contAll = s_cont;
    bad = s_bad != 0;
    if (!chainMode && contAll < (uint32_t)((end - start) / 8)) bad = true;   // ad-hoc threshold (:518)
    if (bad || n == 0 || n >= KNZ_UTF_MAX_SYMS || 3 * (int)n + 6 >= maxTarget) { if (lane == 0) { a.ok[b] = 0; a.out_len[b] = 0; } break; }
    // where the walk stops: behind the last code point that starts in front of count - 4
    int lastP = lastLead;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const int o = (int)wave_shfl((uint32_t)lastP, lane ^ d); lastP = o > lastP ? o : lastP; }
    const int srcStop = lastP + (int)knz_utf_size(src[lastP]);
    if (lane == 0) s_stop = srcStop;
    // ---- pass 2: ranks by (frequency, code point), increasing (:186-193); symbol of sort position r gets index n - 1 - r ------
    unsigned long long est = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += 64) {
        const uint32_t j = j0 + (uint32_t)lane;
        const uint32_t sj = j < n ? syms[j] : 0u;
        const uint32_t fj = j < n ? (uint32_t)map[sj] : 0u;
        const uint64_t kj = ((uint64_t)fj << 32) | sj;
        uint32_t r = 0;
        for (uint32_t k0 = 0; k0 < n; k0 += 1024) {                      // the keys pass through LDS 1024 at a time
            const uint32_t m = min(1024u, n - k0);
            wave_sync_lds();
            for (uint32_t k = lane; k < m; k += 64) { const uint32_t sk = syms[k0 + k]; s_keys[k] = ((uint64_t)(uint32_t)map[sk] << 32) | sk; }
            wave_sync();
            for (uint32_t k = 0; k < m; k++) r += s_keys[k] < kj ? 1u : 0u;
        }
        if (j < n) {
            const uint32_t i = n - 1 - r;
            ranks[j] = i;
            est += (unsigned long long)fj * (i < 128 ? 1u : 2u);
            uint8_t* m = dst + 4 + 3 * (size_t)i;                       // the map (:206-211)
            m[0] = (uint8_t)(sj >> 16); m[1] = (uint8_t)(sj >> 8); m[2] = (uint8_t)sj;
        }
    }
    {
        const uint32_t lo = wave_reduce_add((uint32_t)est), hi = wave_reduce_add((uint32_t)(est >> 32));   // (lo cannot carry out: < 2^31 bytes)
        est = ((unsigned long long)hi << 32) + lo;
    }
    if (est + 10 >= (unsigned long long)maxTarget) { if (lane == 0) { a.ok[b] = 0; a.out_len[b] = 0; } break; }   // estimate (:200,:212-223)
    wave_sync();
    for (uint32_t j = lane; j < n; j += 64) {                            // aliases replace the frequencies (:214-219)
        const uint32_t i = ranks[j];
        map[syms[j]] = i < 128 ? (int32_t)i : (int32_t)(0x10080u | ((i << 1) & 0xFF00u) | (i & 0x7Fu));
    }
    wave_sync();
    __threadfence();
    if (lane == 0) s_go = 1;
    } while (0);
    __syncthreads();
    if (!s_go) return;
    // ---- pass 3: emission, all 16 waves (round 4: one wave walked the block 64 positions at a time, 6.5 of the stage's 8.6 ms) ----------------
    // A tile is 16 K positions, 1 K per wave in 16 rows of 64: the aliases and their offsets inside the wave's stretch stay in registers, the
    // 16 stretch sizes meet in LDS once per tile.
    const uint32_t n = s_n;
    const int srcStop = s_stop;
    const int wave = tid >> 6;
    int dstIdx = 4 + 3 * (int)n;
    if (tid < start) dst[dstIdx + tid] = src[tid];
    dstIdx += start;
    for (int t0 = start; t0 < end; t0 += 16 * KNZ_UTF_FWD_THREADS) {
        uint32_t al[16], off[16];
        uint32_t wsum = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int p = t0 + wave * 1024 + r * 64 + lane;
            uint32_t w = 0, alias = 0;
            if (p < end) {
                const uint32_t b0 = src[p];
                const uint32_t s = knz_utf_size(b0);
                const bool isSym = chainMode ? (((cbits[p >> 3] >> (p & 7)) & 1) != 0) : s != 0;
                if (isSym) {
                    alias = (uint32_t)map[knz_utf_pack(s, b0, src[p + 1], src[p + 2], src[p + 3])];
                    w = 1 + (alias >> 16);
                }
            }
            const uint32_t incl = wave_scan_incl(w);
            al[r] = (alias & 0xFFFFu) | (w << 16);
            off[r] = wsum + incl - w;
            wsum += wave_bcast(incl, 63);
        }
        if (lane == 0) s_wt[wave] = wsum;
        __syncthreads();
        uint32_t base = 0, tot = 0;
        for (int k = 0; k < KNZ_UTF_FWD_THREADS / 64; k++) { const uint32_t v = s_wt[k]; base += k < wave ? v : 0u; tot += v; }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t w = al[r] >> 16;
            if (w) { uint8_t* o = dst + dstIdx + base + off[r]; o[0] = (uint8_t)al[r]; if (w == 2) o[1] = (uint8_t)(al[r] >> 8); }
        }
        dstIdx += (int)tot;
        __syncthreads();
    }
    if (tid >= 64) return;
    if (lane == 0) { dst[0] = (uint8_t)start; dst[1] = (uint8_t)(srcStop - end); dst[2] = (uint8_t)(n >> 8); dst[3] = (uint8_t)n; }
    const int tail = count - srcStop;                                    // last (possibly truncated) bytes (:255-259)
    if (lane < tail) dst[dstIdx + lane] = src[srcStop + lane];
    dstIdx += tail;
    const bool skip = dstIdx >= maxTarget;
    if (lane == 0) {
        a.ok[b] = skip ? 0 : 1; a.out_len[b] = skip ? 0 : (uint32_t)dstIdx;
        if (!skip && a.blk_dt) a.blk_dt[b] = KNZ_DT_UTF8;                          // ctx["dataType"] = DT_UTF8 (:131-133: set once the block validated)
    }
}

__global__ __launch_bounds__(64) void knz_utf_inverse_kernel(UtfArgs a) {
    __shared__ int s_err;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const long long count = (long long)a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const long long cap = (long long)a.out_cap;
    if (count == 0) { if (lane == 0) { a.ok[b] = 1; a.out_len[b] = 0; } return; }
    bool bad = count < 4;
    int start = 0, adjust = 0, n = 0;
    if (!bad) {
        start = src[0] & 3; adjust = src[1] & 3; n = ((int)src[2] << 8) + src[3];
        if (n == 0 || n >= KNZ_UTF_MAX_SYMS || 4 + 3 * (long long)n > count) bad = true;        // :289-291
    }
    if (bad) { if (lane == 0) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; a.out_len[b] = 0; } return; }
    uint64_t* inv = a.inv_map + (size_t)b * KNZ_UTF_MAX_SYMS;
    if (lane == 0) s_err = 0;
    wave_sync();
    for (int i = lane; i < KNZ_UTF_MAX_SYMS; i += 64) {                  // unpackUTF1 (:578-609); entries past n have length 0
        uint64_t e = 0;
        if (i < n) {
            const uint8_t* q = src + 4 + 3 * (size_t)i;
            const uint32_t in = ((uint32_t)q[0] << 16) | ((uint32_t)q[1] << 8) | q[2];
            const uint32_t sz = in >> 19;
            uint32_t v = 0, len = 0;
            if (sz == 0) { v = in & 0xFF; len = 1; }
            else if (sz == 1) { v = ((in >> 8) & 0xFF) | ((in & 0xFF) << 8); len = 2; }
            else if (sz == 2) { v = (((in >> 12) & 0x0F) | 0xE0) | ((((in >> 6) & 0x3F) | 0x80) << 8) | (((in & 0x3F) | 0x80) << 16); len = 3; }
            else if (sz >= 4) { v = (((in >> 18) & 0x07) | 0xF0) | ((((in >> 12) & 0x3F) | 0x80) << 8) | ((((in >> 6) & 0x3F) | 0x80) << 16) | (((in & 0x3F) | 0x80) << 24); len = 4; }
            else s_err = 1;                                              // sz == 3: invalid alias (:322-324)
            e = (uint64_t)v | ((uint64_t)len << 32);
        }
        inv[i] = e;
    }
    wave_sync();
    __threadfence();
    long long srcIdx = 4 + 3 * (long long)n;
    const long long srcEnd = count - 4 + adjust;
    const long long dstEnd = cap - 4;
    if (s_err || dstEnd < 0 || srcEnd < srcIdx || srcEnd > count || srcIdx + start > count) { if (lane == 0) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; a.out_len[b] = 0; } return; }
    long long dstIdx = 0;
    if (lane < start) dst[lane] = src[srcIdx + lane];
    srcIdx += start; dstIdx += start;
    // aliases: state 0 = at an alias, 1 = second byte of a two-byte alias. f = (next state from 0) | (next state from 1) << 1
    uint32_t carry = 0;                                                  // state in front of the row
    bool fail = srcIdx > srcEnd;
    for (long long p0 = srcIdx; p0 < srcEnd && !fail; p0 += 64) {
        const long long p = p0 + lane;
        const bool in = p < srcEnd;
        const uint32_t b0 = in ? src[p] : 0u;
        // inclusive scan of the composed transition functions (state -> state after my byte); bytes behind srcEnd are identity
        uint32_t f = in ? (b0 >= 128 ? 1u : 0u) : 2u;                    // from 0: to (b0 >= 128), from 1: to 0 ; identity = 0b10
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t g = wave_shfl(f, lane - d);                   // the function of the lanes in front of me
            if (lane >= d) f = ((f >> (g & 1)) & 1) | (((f >> ((g >> 1) & 1)) & 1) << 1);   // f o g
        }
        const uint32_t fprev = wave_shfl(f, lane - 1);
        const uint32_t stIn = lane == 0 ? carry : ((fprev >> carry) & 1);   // my state before my byte
        const uint32_t stLast = (wave_bcast(f, 63) >> carry) & 1;
        const bool isAlias = in && stIn == 0;
        uint32_t alias = b0, len = 0;
        uint64_t e = 0;
        if (isAlias) {
            if (b0 >= 128) {
                if (p + 1 >= srcEnd) fail = true;                        // :353-355
                else alias = ((uint32_t)src[p + 1] << 7) + (b0 & 0x7F);
            }
            if (!fail) { e = inv[alias]; len = (uint32_t)(e >> 32); }
        }
        const uint32_t incl = wave_scan_incl(len);
        const long long at = dstIdx + (incl - len);
        if (isAlias && !fail) {
            if (at >= dstEnd) fail = true;                               // the loop stops at dstEnd with input left (:347, :366)
            else for (uint32_t q = 0; q < len; q++) dst[at + q] = (uint8_t)(e >> (8 * q));
        }
        if (wave_ballot(fail) != 0) fail = true;
        dstIdx += (long long)wave_bcast(incl, 63);
        carry = stLast;
    }
    if (!fail && carry != 0) fail = true;                                // a two-byte alias cut by srcEnd
    if (!fail && dstIdx > cap - count + srcEnd) fail = true;             // :366
    if (fail) { if (lane == 0) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; a.out_len[b] = 0; } return; }
    const long long tail = count - srcEnd;                               // <= 4
    if (lane < tail) dst[dstIdx + lane] = src[srcEnd + lane];
    dstIdx += tail;
    if (lane == 0) { a.ok[b] = 1; a.out_len[b] = (uint32_t)dstIdx; }
}

