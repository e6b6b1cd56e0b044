// Wave-parallel pieces of the Huffman decode path (gfx950). The serial forms in huffman_dec.hip stay as the reference-exact
// fallback for anything unusual (non-canonical Exp-Golomb encodings a Go encoder never writes, inconsistent fragments).
//
//   knz_huf_parse_header_wave   the chunk header (alphabet + signed Exp-Golomb code-length deltas + 4 varints,
//                               HuffmanCodec.go:620-657, EntropyUtils.go:71-119, ExpGolombCodec.go:159-190) parsed by all 64
//                               lanes: lane l owns 32 bits of the delta section, assumes a code boundary at its first bit,
//                               and the lanes hand each other their exit offsets until nothing moves (Exp-Golomb codes
//                               re-synchronise within a few bits, so this takes 2-3 rounds instead of 256 serial codes).
//   knz_huf_decode_par_kernel   one wave per 16 KiB chunk; every fragment bit string is cut into 16 equal sub-ranges, one per
//                               lane (decodeChunkV6 :832-969 runs 4 serial decoders). A lane starts on a guessed boundary,
//                               decodes to the end of its sub-range and passes the bit position it stopped at to its
//                               neighbour; Huffman codes self-synchronise, so after one more round every lane starts on a
//                               true code boundary (verified: the hand-over positions must be a fixed point, the symbol
//                               counts must add up to the fragment length, the last lane must end on the last bit).
#include "bits.h"

#ifdef KNZ_PROFILE_PHASES
__device__ unsigned long long g_knz_prof[32];
#define KNZ_PROF_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define KNZ_PROF_ADD(i, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_knz_prof[i], (b) - (a)); } while (0)
#define KNZ_PROF_INC(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_knz_prof[i], (unsigned long long)(v)); } while (0)
#else
#define KNZ_PROF_INC(i, v)
#define KNZ_PROF_T(var)
#define KNZ_PROF_ADD(i, a, b)
#endif

#define KNZ_HUF_SYNC_BITS 192        // look-back of the boundary search: ~20 codes, Huffman codes re-synchronise well within
#define KNZ_HUF_LANE_CAP 84           // symbols a lane keeps from its decode pass (sub-ranges hold ~64); more -> a write pass.
                                      // 21 dwords: an odd row stride spreads the 64 rows of a wave over all 32 LDS banks
#define KNZ_HW_WORDS 136              // header window: 128 words + padding for the 64-bit look-ahead

struct KnzHufHdr {
    uint32_t count;                   // alphabet size
    uint32_t fb[4];                   // fragment bit counts (count > 1)
    uint32_t end;                     // bit offset (inside the window) just past the header
    uint32_t status;                  // 0 ok, 1 unusual encoding -> serial parser, 2 invalid (ERR_PROCESS_BLOCK)
};

// Header windows: BE words (already byte-swapped) held in LDS, either a plain array or a stretch of the walker's ring.
struct KnzLinWin {
    const uint32_t* p;
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return p[i]; }
};
struct KnzRingWin {
    const uint32_t* ring; uint32_t base, mask;
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return ring[(base + i) & mask]; }
};

// 32 bits at bit offset `bit` of a window
template <class Win>
__device__ __forceinline__ uint32_t knz_win32(const Win& w, uint32_t bit) {
    const uint32_t i = bit >> 5, o = bit & 31;
    const uint64_t v = ((uint64_t)w.at(i) << 32) | w.at(i + 1);
    return (uint32_t)((v << o) >> 32);
}

// bits [pos, pos + 32) of the 64-bit window {hi, lo}, pos in 0..31
__device__ __forceinline__ uint32_t knz_top32(uint32_t hi, uint32_t lo, uint32_t pos) {
    return pos ? ((hi << pos) | (lo >> (32 - pos))) : hi;
}

// The signed Exp-Golomb code-length deltas of a chunk header (readLengths, HuffmanCodec.go:620-657; ExpGolombCodec.go:159-190),
// `count` codes from bit e0 of the window. Lane l owns the codes that START in bits [p0, p0 + W) of the section, p0 = e0 + W*l.
// It assumes a code boundary at its entry offset, walks to the first boundary at or past p0 + W (a run of '1' = delta 0 codes
// and at most one longer code per step, no branches) and hands that offset to its neighbour, until nothing moves: lane 0 is
// exact, and these codes re-synchronise within a few bits, so this takes 2-3 rounds instead of `count` serial steps.
// Returns 0 and the bit offset just past the section, 1 when the serial parser has to take over (a delta a Go encoder never
// writes, or a section longer than 64 * W bits), 2 for an invalid length.
template <bool WANT_LEN, int W, class Win>
__device__ __forceinline__ uint32_t knz_huf_delta_section(const Win& w, uint32_t e0, uint32_t count, uint8_t* s_alpha, uint8_t* s_len, int lane, uint32_t& endOut) {
    const uint32_t p0 = e0 + (uint32_t)W * (uint32_t)lane;
    const uint32_t whi = knz_win32(w, p0), wlo = knz_win32(w, p0 + 32);
    const uint64_t win = ((uint64_t)whi << 32) | wlo;
    const bool relevant = (uint32_t)lane <= (count * 8 + W - 1) / W;     // a usual code has at most 8 bits
    uint32_t o = 0, c = 0, ex = 0;
    int ds = 0;
    bool odd = false, need = relevant;                                   // lanes behind the section carry nothing
    for (int rounds = 0; ; rounds++) {
        // only the lanes whose entry offset moved walk again: the last round (nothing moved) costs one shuffle and one ballot
        uint32_t pos = need ? o : (uint32_t)W;
        if (need) { c = 0; ds = 0; odd = false; }
        while (pos < (uint32_t)W) {
            const uint32_t bits = (uint32_t)((win << pos) >> 32);
            const uint32_t ones = min((uint32_t)__builtin_clz(~bits | 1u), (uint32_t)W - pos);
            const uint32_t pos1 = pos + ones;
            const uint32_t b2 = (uint32_t)((win << pos1) >> 32);             // (pos1 <= W <= 32)
            const bool lng = pos1 < (uint32_t)W && !(b2 >> 31);            // a code with leading zeros starts inside my bits
            const uint32_t z = (uint32_t)__builtin_clz(b2 | 1u);
            if (lng && z > 3) odd = true;                                  // not a length delta a Go encoder writes
            if (WANT_LEN) {
                const uint32_t zz = z & 3;
                const uint32_t val = (b2 << (zz + 1)) >> (31 - zz);       // the z+1 bits behind the terminator
                const int mag = (int)((val >> 1) + (1u << zz) - 1u);
                ds += lng ? ((val & 1) ? -mag : mag) : 0;
            }
            pos = odd ? (uint32_t)W : pos1 + (lng ? 2 * z + 2 : 0u);
            c += ones + ((lng && !odd) ? 1u : 0u);
            ex = pos - (uint32_t)W;
        }
        uint32_t no = wave_shfl(ex, lane - 1);
        if (lane == 0) no = 0;
        need = relevant && no != o;
        o = no;
#ifdef KNZ_EMU_STATS
        if (lane == 0) { extern unsigned long long g_stat[8]; g_stat[1]++; }
#endif
        if (wave_ballot(need) == 0) break;
        if (rounds > 66) return 1;
    }
    const uint32_t cincl = wave_scan_incl(c);
    const uint32_t idx0 = cincl - c;
    // the lane holding code count-1; every lane before it must be made of usual codes
    const uint64_t em = wave_ballot(idx0 < count && idx0 + c >= count);
    const uint64_t om = wave_ballot(odd);
    if (em == 0) return 1;                                               // section longer than 64 x W bits
    const uint32_t endLane = (uint32_t)(__ffsll((unsigned long long)em) - 1);
    // (an unusual pattern in the end lane itself lies behind the section: its codes are counted up to that point only)
    if (om != 0 && (uint32_t)(__ffsll((unsigned long long)om) - 1) < endLane) return 1;
    uint32_t myEnd = 0;
    if ((uint32_t)lane == endLane) {
        uint32_t pos = o, idx = idx0;
        while (idx < count) {
            const uint32_t bits = (uint32_t)((win << pos) >> 32);
            const uint32_t ones = (uint32_t)__builtin_clz(~bits | 1u);
            if (ones) { const uint32_t take = min(ones, count - idx); idx += take; pos += take; continue; }
            pos += 2u * (uint32_t)__builtin_clz(bits | 1u) + 2u;
            idx++;
        }
        myEnd = p0 + pos;
    }
    endOut = wave_readlane(myEnd, endLane);
    if (WANT_LEN) {
        wave_sync_lds();                                                 // s_alpha
        const int dincl = (int)wave_scan_incl((uint32_t)ds);
        int cur = 2 + dincl - ds;
        bool bad = false;
        uint32_t pos = (uint32_t)lane <= endLane ? o : (uint32_t)W, idx = idx0;
        while (pos < (uint32_t)W && idx < count) {
            const uint32_t bits = (uint32_t)((win << pos) >> 32);
            if (bits >> 31) pos += 1;
            else {
                const uint32_t z = (uint32_t)__builtin_clz(bits | 1u);
                const uint32_t val = (bits << (z + 1)) >> (31 - z);
                const int mag = (int)((val >> 1) + (1u << z) - 1u);
                cur += (val & 1) ? -mag : mag;
                pos += 2 * z + 2;
            }
            if (cur <= 0 || cur > KNZ_HUF_MAXLEN) { bad = true; break; }     // readLengths :640-645
            s_len[s_alpha[idx]] = (uint8_t)cur;
            idx++;
        }
        if (wave_ballot(bad) != 0) return 2;
    }
    return 0;
}

// w: window of BE words (LDS), h0: bit offset of the chunk header inside it (< 32). All 64 lanes
// of ONE wave call it; the result is wave-uniform. WANT_LEN: also fill s_alpha[0..count) and s_len[symbol] (zeroed by the
// caller).
template <bool WANT_LEN, class Win>
__device__ __forceinline__ KnzHufHdr knz_huf_parse_header_wave(const Win& w, uint32_t h0, uint8_t* s_alpha, uint8_t* s_len, int lane) {
    KNZ_PROF_T(pA);
    KnzHufHdr h;
    h.count = 0; h.end = 0; h.status = 0;
    h.fb[0] = h.fb[1] = h.fb[2] = h.fb[3] = 0;
    // ---- alphabet (EntropyUtils.go:71-119) ------------------------------------------------------------------------------
    const uint32_t b0 = wave_uniform(knz_win32(w, h0));
    uint32_t e0, count;
    if ((b0 >> 31) == 0) {
        if ((b0 >> 30) & 1) { h.status = 2; return h; }                 // empty alphabet
        count = 256;
        e0 = h0 + 2;
        if (WANT_LEN) {
#pragma unroll
            for (int k = 0; k < 4; k++) s_alpha[64 * k + lane] = (uint8_t)(64 * k + lane);
        }
    } else {
        const uint32_t lastMask = (b0 >> 26) & 31;
        e0 = h0 + 6 + 8 * (lastMask + 1);
        const uint32_t byte = (uint32_t)lane <= lastMask ? (knz_win32(w, h0 + 6 + 8 * (uint32_t)lane) >> 24) : 0u;
        const uint32_t pc = (uint32_t)__popc(byte);
        const uint32_t incl = wave_scan_incl(pc);
        count = wave_bcast(incl, 63);
        if (WANT_LEN) {
            uint32_t idx = incl - pc;
            for (int j = 0; j < 8; j++) if ((byte >> j) & 1) s_alpha[idx++] = (uint8_t)(8 * lane + j);
        }
        if (count == 0) { h.status = 2; return h; }
    }
    h.count = count;
    KNZ_PROF_T(p0t);
    // ---- code-length deltas: 16 bits of the section per lane when that can cover it, 32 otherwise ----------------------------
    uint32_t end = 0;
    uint32_t est = 1;
    if (count <= 192) est = knz_huf_delta_section<WANT_LEN, 16, Win>(w, e0, count, s_alpha, s_len, lane, end);
    if (est == 1) est = knz_huf_delta_section<WANT_LEN, 32, Win>(w, e0, count, s_alpha, s_len, lane, end);   // (section longer than 64 x 16 bits)
    if (est != 0) { h.status = est; return h; }
    KNZ_PROF_T(p3);
    h.end = end;
    // ---- fragment sizes: 4 varints (EntropyUtils.go:278-296), bytes fetched by 20 lanes, assembled on the scalar unit ------
    if (count > 1) {
        // continuation bits of the next 20 bytes -> the 4 lengths on the scalar unit; lane j then assembles varint j
        const uint32_t vb = lane < 20 ? (knz_win32(w, end + 8u * (uint32_t)lane) >> 24) : 0u;
        const uint32_t cm = (uint32_t)wave_ballot(vb >= 128);
        uint32_t at[5];
        at[0] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t rest = ~(cm >> at[j]);
            at[j + 1] = at[j] + min(5u, (uint32_t)__ffs((int)rest));          // bytes up to and including the first one < 128
        }
        const uint32_t myAt = lane == 0 ? at[0] : (lane == 1 ? at[1] : (lane == 2 ? at[2] : at[3]));
        const uint32_t myLen = (lane == 0 ? at[1] : (lane == 1 ? at[2] : (lane == 2 ? at[3] : at[4]))) - myAt;
        const uint32_t q0 = knz_win32(w, end + 8 * myAt), q1 = knz_win32(w, end + 8 * myAt + 32);
        uint32_t v = (q0 >> 24) & 0x7F;
        if (myLen > 1) v |= ((q0 >> 16) & 0x7F) << 7;
        if (myLen > 2) v |= ((q0 >> 8) & 0x7F) << 14;
        if (myLen > 3) v |= (q0 & 0x7F) << 21;
        if (myLen > 4) v |= ((q1 >> 24) & 0x0F) << 28;
        h.fb[0] = wave_bcast(v, 0); h.fb[1] = wave_bcast(v, 1); h.fb[2] = wave_bcast(v, 2); h.fb[3] = wave_bcast(v, 3);
        h.end = end + 8 * at[4];
    }
    KNZ_PROF_T(p4);
    KNZ_PROF_ADD(17, pA, p0t); KNZ_PROF_ADD(18, p0t, p3); KNZ_PROF_ADD(21, p3, p4); KNZ_PROF_INC(22, 1); KNZ_PROF_INC(23, count);
    return h;
}

// Bit reader of one lane over a fragment sub-range: MSB-first, 64-bit window, words fetched THREE refills ahead (a refill
// comes every ~6 symbols; one word of look-ahead left the s_waitcnt in front of an L2 round trip, SQ_WAIT_ANY was 64 % of the
// wave cycles). Word indexes are relative to the chunk's first word and clamped to the stream end instead of tested.
struct KnzFragReader {
    const uint32_t* w;
    uint32_t maxRel, next, navail, p0, p1, p2;
    uint64_t win;
    __device__ __forceinline__ uint32_t ld(uint32_t i) const { return w[min(i, maxRel)]; }
    __device__ __forceinline__ void init(const uint32_t* chunkWords, uint32_t maxRelWord, uint32_t relbit) {
        w = chunkWords; maxRel = maxRelWord;
        const uint32_t q = relbit >> 5, off = relbit & 31;
        const uint32_t a = ld(q), b = ld(q + 1);
        p0 = ld(q + 2); p1 = ld(q + 3); p2 = ld(q + 4);
        win = (((uint64_t)knz_bswap32(a) << 32) | knz_bswap32(b)) << off;
        navail = 64 - off;
        next = q + 5;
    }
    __device__ __forceinline__ uint32_t peek12() {
        if (navail <= 32) { win |= (uint64_t)knz_bswap32(p0) << (32 - navail); navail += 32; p0 = p1; p1 = p2; p2 = ld(next); next++; }
        return (uint32_t)(win >> 52);
    }
    __device__ __forceinline__ void consume(uint32_t n) { win <<= n; navail -= n; }
};

// ---------------------------------------------------------------------------------------------------------------------
// One 256-thread workgroup per 16 KiB chunk: wave 0 parses the header, all threads build the code table, then wave j owns
// fragment j and lane s its sub-range s of 64.
// LDS of one chunk decode (a struct so that the fused walk+decode kernel can overlay it with the walker's ring)
struct __attribute__((aligned(16))) KnzHufParShared {
    uint16_t s_table[1 << KNZ_HUF_MAXLEN];
    uint8_t s_outb[256 * KNZ_HUF_LANE_CAP];   // >= KNZ_HUF_CHUNK: lane buffers, then the chunk (16-byte aligned: 8192 bytes in)
    uint32_t s_hw[KNZ_HW_WORDS];
    uint8_t s_len[256];
    uint8_t s_alpha[256];
    uint16_t s_C[258];
    uint8_t s_symAt[256];
    uint32_t s_cnt[4][16];
    KnzHufHdr s_hdr;
    int s_flag, s_over;
};

// chunk `slot` (= block * chunks_per_block + chunk) whose first bit is `cbit`, by the 256 threads of a workgroup
__device__ __forceinline__ void knz_huf_decode_par_body(const HufDecArgs& a, uint8_t* fallback, const uint32_t slot, const uint64_t cbit, KnzHufParShared& sh) {
    uint16_t (&s_table)[1 << KNZ_HUF_MAXLEN] = sh.s_table;
    uint8_t (&s_outb)[256 * KNZ_HUF_LANE_CAP] = sh.s_outb;
    uint32_t (&s_hw)[KNZ_HW_WORDS] = sh.s_hw;
    uint8_t (&s_len)[256] = sh.s_len;
    uint8_t (&s_alpha)[256] = sh.s_alpha;
    uint16_t (&s_C)[258] = sh.s_C;
    uint8_t (&s_symAt)[256] = sh.s_symAt;
    uint32_t (&s_cnt)[4][16] = sh.s_cnt;
    KnzHufHdr& s_hdr = sh.s_hdr;
    int& s_flag = sh.s_flag;
    int& s_over = sh.s_over;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cpb = a.chunks_per_block;
    const uint32_t b = slot / cpb, k = slot % cpb;
    if (tid == 0) fallback[slot] = 0;
    const uint32_t preLen = a.blk_pre_len[b];
    if (a.blk_status[b] != 0) return;
    if ((uint64_t)k * KNZ_HUF_CHUNK >= preLen) return;
    const uint32_t n = min((uint32_t)KNZ_HUF_CHUNK, preLen - k * KNZ_HUF_CHUNK);
    uint8_t* dst = a.out + a.blk_out_off[b] + (size_t)k * KNZ_HUF_CHUNK;
    uint32_t entropy = a.entropy;
    if (a.blk_mode[b] & 0x80) entropy = KNZ_E_NONE;
    const uint64_t limit = a.nbytes << 3;

    if (entropy == KNZ_E_NONE || n < 32) {   // raw copy at an arbitrary bit offset
        for (uint32_t i = tid * 4; i < n; i += 1024) {
            uint32_t w = knz_fetch32(a.stream, (int64_t)(cbit + 8ull * i), (int64_t)limit);
            for (uint32_t j = 0; j < 4 && i + j < n; j++) dst[i + j] = (uint8_t)(w >> (24 - 8 * j));
        }
        return;
    }

    KNZ_PROF_T(t0);
    // ---- header ---------------------------------------------------------------------------------------------------------
    {
        const uint32_t* words = (const uint32_t*)a.stream;
        const uint64_t nwords = (a.nbytes + 3) >> 2, w0i = cbit >> 5;
        if (tid < KNZ_HW_WORDS) s_hw[tid] = (w0i + tid < nwords) ? knz_bswap32(words[w0i + tid]) : 0u;
        s_len[tid] = 0;
        if (tid == 0) { s_flag = 0; s_over = 0; }
    }
    __syncthreads();
    if (wave == 0) {
        const KnzHufHdr hh = knz_huf_parse_header_wave<true>(KnzLinWin{s_hw}, (uint32_t)(cbit & 31), s_alpha, s_len, lane);
        if (lane == 0) s_hdr = hh;
    }
    __syncthreads();
    const KnzHufHdr hdr = s_hdr;
    if (hdr.status == 1) { if (tid == 0) fallback[slot] = 1; return; }
    if (hdr.status == 2) { if (tid == 0) a.blk_status[b] = KNZ_ERR_PROCESS_BLOCK; return; }
    KNZ_PROF_T(t1);
    KNZ_PROF_ADD(0, t0, t1);
    const uint32_t count = hdr.count;
    if (count == 1) {                          // decodeV6 :778-786
        const uint8_t v = s_alpha[0];
        for (uint32_t i = tid; i < n; i += 256) dst[i] = v;
        return;
    }

    // ---- canonical codes (generateCanonicalCodes :37-77): thread = symbol, codes are handed out by (length, symbol) ---------
    const uint32_t myLen = s_len[tid];
    uint32_t myBefore = 0;
    {
        const uint64_t below = ((uint64_t)1 << lane) - 1;
        for (uint32_t L = 1; L <= KNZ_HUF_MAXLEN; L++) {
            const uint64_t m = wave_ballot(myLen == L);
            if (lane == 0) s_cnt[wave][L] = (uint32_t)__popcll(m);
            if (myLen == L) myBefore = (uint32_t)__popcll(m & below);
        }
    }
    __syncthreads();
    uint32_t full = 0, myC = 0, myRank = 0, rankAcc = 0;
    for (uint32_t L = 1; L <= KNZ_HUF_MAXLEN; L++) {
        const uint32_t c0 = s_cnt[0][L], c1 = s_cnt[1][L], c2 = s_cnt[2][L], c3 = s_cnt[3][L];
        if (L == myLen) {
            const uint32_t prior = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u) + myBefore;
            myC = full + (prior << (KNZ_HUF_MAXLEN - L));
            myRank = rankAcc + prior;
        }
        const uint32_t tot = c0 + c1 + c2 + c3;
        full += tot << (KNZ_HUF_MAXLEN - L);
        rankAcc += tot;
    }
    if (full > (1u << KNZ_HUF_MAXLEN)) { if (tid == 0) a.blk_status[b] = KNZ_ERR_PROCESS_BLOCK; return; }   // buildDecodingTable :683-685
    if (myLen) { s_C[myRank] = (uint16_t)myC; s_symAt[myRank] = (uint8_t)tid; }
    if (tid == 0) s_C[count] = (uint16_t)full;                          // <= 4096
    __syncthreads();
    // ---- 4096-entry table (buildDecodingTable :661-697): thread t fills entries [16t, 16t+16) -------------------------------
    {
        const uint32_t e0 = 16u * (uint32_t)tid;
        uint32_t lo = 0, hi = count;                                    // last rank with C <= e0
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_C[mid] <= e0) lo = mid; else hi = mid; }
        uint32_t r = lo;
        uint32_t nextC = s_C[r + 1];
        uint32_t sym = s_symAt[r];
        uint32_t val = (sym << 8) | s_len[sym];
        uint32_t pair = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16; i++) {
            const uint32_t e = e0 + i;
            while (e >= nextC && r + 1 < count) { r++; nextC = s_C[r + 1]; sym = s_symAt[r]; val = (sym << 8) | s_len[sym]; }
            const uint32_t v = e < full ? val : 7u;                       // table[i] = 7 where no code ends (:665-667)
            if (i & 1) ((uint32_t*)s_table)[(e0 + i) >> 1] = pair | (v << 16); else pair = v;
        }
    }
    __syncthreads();
    KNZ_PROF_T(t2);
    KNZ_PROF_ADD(1, t1, t2);

    // ---- fragments: wave = fragment j, lane = sub-range s ----------------------------------------------------------------
    const uint32_t F = n >> 2;
    const uint32_t j = (uint32_t)wave, s = (uint32_t)lane;
    const uint64_t hdrEnd = ((cbit >> 5) << 5) + hdr.end;
    const uint32_t fbj = j == 0 ? hdr.fb[0] : (j == 1 ? hdr.fb[1] : (j == 2 ? hdr.fb[2] : hdr.fb[3]));
    const uint64_t fp = hdrEnd + (j > 0 ? hdr.fb[0] : 0u) + (uint64_t)(j > 1 ? hdr.fb[1] : 0u) + (uint64_t)(j > 2 ? hdr.fb[2] : 0u);
    const uint64_t tailpos = hdrEnd + (uint64_t)hdr.fb[0] + hdr.fb[1] + hdr.fb[2] + hdr.fb[3];
    bool sane = (int32_t)(hdr.fb[0] | hdr.fb[1] | hdr.fb[2] | hdr.fb[3]) >= 0 && tailpos + 8ull * (n & 3) <= limit + 7;
    sane = sane && fbj <= 12u * F;                                      // a symbol costs at most 12 bits
    uint32_t st = 0, nsym = 0, e = 0, off = 0;
    // fragment positions relative to the chunk's first word (< 2^18 bits once `sane` holds)
    const uint32_t* cwords = (const uint32_t*)a.stream + (cbit >> 5);
    const uint64_t nwordsAll = (a.nbytes + 3) >> 2;
    const uint32_t maxRel = (cbit >> 5) < nwordsAll ? (uint32_t)min((uint64_t)0x7FFFFFFFu, nwordsAll - 1 - (cbit >> 5)) : 0u;
    const uint32_t fpRel = (uint32_t)(fp - ((cbit >> 5) << 5));
    if (sane) {
        const uint32_t B = (fbj + 63) >> 6;
        const uint32_t E = min(fbj, (s + 1) * B);
        // 1) synchronisation: start KNZ_HUF_SYNC_BITS before my sub-range on a guessed boundary and stop on the first code
        //    boundary inside it (a symbol belongs to the lane in whose sub-range it starts, so this is where my neighbour ends)
        const uint32_t R = min(fbj, s * B);
        st = R;
        if (s > 0 && R < fbj) {
            uint32_t pos = R > KNZ_HUF_SYNC_BITS ? R - KNZ_HUF_SYNC_BITS : 0u;
            KnzFragReader r;
            r.init(cwords, maxRel, fpRel + pos);
            while (pos < R) {
                const uint32_t val = s_table[r.peek12()];
                r.consume(val & 0xFF);
                pos += val & 0xFF;
            }
            st = pos;
        }
        // 2) decode my sub-range from there; the place I stop must be where my neighbour started, otherwise it restarts there
        bool redo = true;
        for (int round = 0; ; round++) {
            if (redo) {
                KnzFragReader r;
                r.init(cwords, maxRel, fpRel + st);
                uint32_t pos = st, cnt = 0;
                uint8_t* mine = s_outb + (size_t)tid * KNZ_HUF_LANE_CAP;       // the symbols stay in my LDS row (clamped)
                while (pos < E) {
                    const uint32_t val = s_table[r.peek12()];
                    r.consume(val & 0xFF);
                    pos += val & 0xFF;
                    mine[min(cnt, (uint32_t)KNZ_HUF_LANE_CAP - 1)] = (uint8_t)(val >> 8);
                    cnt++;
                }
                nsym = cnt; e = pos;
            }
            uint32_t ns = wave_shfl(e, lane - 1);
            if (s == 0) ns = 0;
            redo = ns != st;
            st = ns;
#ifdef KNZ_EMU_STATS
            if (tid == 0) { extern unsigned long long g_stat[8]; g_stat[0]++; }
#endif
            if (wave_ballot(redo) == 0) break;
            if (round > 66) { sane = false; break; }
        }
        // symbol offsets inside the fragment; consistency with the count-driven reference decoder
        const uint32_t incl = wave_scan_incl(nsym);
        off = incl - nsym;
        if (s == 63 && !(incl == F && e == fbj)) sane = false;
    }
    if (wave_ballot(!sane) != 0 && lane == 0) s_flag = 1;
    if (wave_ballot(nsym > KNZ_HUF_LANE_CAP) != 0 && lane == 0) s_over = 1;
    __syncthreads();
    KNZ_PROF_T(t3);
    KNZ_PROF_ADD(2, t2, t3);
    if (s_flag) { if (tid == 0) fallback[slot] = 1; return; }
    KNZ_PROF_INC(24, s_over ? 1 : 0);
    if (!s_over) {
        // every lane kept all of its symbols: pull the row into registers, then (the rows and the chunk share the LDS area)
        // store it at its place j*F + off: whole words through a byte funnel, the ragged ends byte by byte
        uint32_t row[KNZ_HUF_LANE_CAP / 4];
#pragma unroll
        for (int m = 0; m < KNZ_HUF_LANE_CAP / 4; m++) row[m] = ((const uint32_t*)(s_outb + (size_t)tid * KNZ_HUF_LANE_CAP))[m];
        __syncthreads();
        const uint32_t d0 = j * F + off, aoff = d0 & 3;
        uint32_t* wout = (uint32_t*)s_outb + (d0 >> 2);
#pragma unroll
        for (int m = 0; m <= KNZ_HUF_LANE_CAP / 4; m++) {
            // destination word m holds my bytes [4m - aoff, 4m - aoff + 4)
            const uint32_t lo = m > 0 ? row[m > 0 ? m - 1 : 0] : 0u, hi = m < KNZ_HUF_LANE_CAP / 4 ? row[m < KNZ_HUF_LANE_CAP / 4 ? m : 0] : 0u;
            const uint32_t v = aoff ? (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (4 - aoff))) : hi;
            const int b0 = 4 * m - (int)aoff;                       // first source byte of this word
            if (b0 >= 0 && b0 + 4 <= (int)nsym) wout[m] = v;
            else {
#pragma unroll
                for (int q = 0; q < 4; q++) if (b0 + q >= 0 && b0 + q < (int)nsym) ((uint8_t*)(wout + m))[q] = (uint8_t)(v >> (8 * q));
            }
        }
    } else {
        __syncthreads();
        KnzFragReader r;
        r.init(cwords, maxRel, fpRel + st);
        uint8_t* o = s_outb + (size_t)j * F + off;
        for (uint32_t i = 0; i < nsym; i++) {
            const uint32_t val = s_table[r.peek12()];
            r.consume(val & 0xFF);
            o[i] = (uint8_t)(val >> 8);
        }
    }
    KNZ_PROF_T(t4);
    KNZ_PROF_ADD(3, t3, t4);
    if ((uint32_t)tid < (n & 3)) s_outb[4 * F + tid] = (uint8_t)(knz_fetch32(a.stream, (int64_t)(tailpos + 8ull * tid), (int64_t)limit) >> 24);
    __syncthreads();
    if ((((uintptr_t)dst) & 15) == 0) {
        for (uint32_t i = tid; i < (n >> 4); i += 256) ((uint4*)dst)[i] = ((const uint4*)s_outb)[i];
        for (uint32_t i = (n & ~15u) + tid; i < n; i += 256) dst[i] = s_outb[i];
    } else {
        for (uint32_t i = tid; i < n; i += 256) dst[i] = s_outb[i];
    }
    KNZ_PROF_T(t5);
    KNZ_PROF_ADD(4, t4, t5);
    KNZ_PROF_ADD(5, t0, t5);
}

__global__ __launch_bounds__(256) void knz_huf_decode_par_kernel(HufDecArgs a, uint8_t* fallback) {
    __shared__ KnzHufParShared sh;
    knz_huf_decode_par_body(a, fallback, blockIdx.x, a.chunk_bit[blockIdx.x], sh);
}

// ---------------------------------------------------------------------------------------------------------------------
// One wave per block, the serial walk runs on lane 0 only: putting several blocks on the lanes of one wave makes
// their data-dependent loops diverge and the wave then pays for the union of all paths.
#define KNZ_WALK_RING 8192            // words of the stream kept ahead of the walk (32 KiB, power of two)
#define KNZ_WALK_AHEAD 6656           // prefetch target: 26 KiB past the current chunk start (a chunk is < 25 KiB)


// 1 KiB of the stream (256 words from word index `base`, a multiple of 256), 16 bytes per lane; words past the end read 0
__device__ __forceinline__ uint4 knz_walk_load_granule(const uint32_t* words, uint64_t nwords, uint64_t base, int lane) {
    const uint64_t i = base + 4 * (uint64_t)lane;
    uint4 v; v.x = v.y = v.z = v.w = 0;
    if (base + 256 <= nwords) v = *(const uint4*)(words + i);           // wave-uniform: the whole granule is inside
    else {
        if (i < nwords) v.x = words[i];
        if (i + 1 < nwords) v.y = words[i + 1];
        if (i + 2 < nwords) v.z = words[i + 2];
        if (i + 3 < nwords) v.w = words[i + 3];
    }
    return v;
}

// Wave 0 walks the block (the chain chunk k -> chunk k+1 is serial by format). Wave 1 is its feeder: it streams the block's
// compressed bytes into the LDS ring ahead of the walk (up to KNZ_WALK_AHEAD words past the chunk being parsed, never over
// words the walker may still read), so that ring traffic costs the chain nothing. The two meet through three LDS words:
// s_sync[0] = stream word index (relative to the ring origin) up to which the ring is filled, [1] = first word the walker
// still needs, [2] = walker finished.
struct __attribute__((aligned(16))) KnzWalkShared {
    uint32_t s_ring[KNZ_WALK_RING];
    uint8_t s_lut[1 << KNZ_EXPG_WIN];
    uint32_t s_sync[4];
};

// chunk_bit values that are not positions (the fused kernel's decoders poll chunk_bit, which is preset to NOT_READY)
#define KNZ_CHUNK_NOT_READY (~0ull)
#define KNZ_CHUNK_ERR (~0ull - 1)

struct KnzBlkHdr { uint32_t mode, skipFlags, preLen, entropy; uint64_t ck; int32_t status; };

// block header (decodingTask.decode, v2/io/CompressedStream.go:1878-1914), wave-uniform
__device__ __forceinline__ KnzBlkHdr knz_walk_block_header(const WalkBlocksArgs& a, KnzWaveReader& r) {
    KnzBlkHdr h;
    h.status = 0; h.mode = 0; h.skipFlags = 0; h.preLen = a.given_len; h.entropy = a.entropy; h.ck = 0;
    if (!a.payload_only) {
        h.mode = r.read(8);
        if (h.mode & 0x80) { h.entropy = KNZ_E_NONE; h.skipFlags = 0xFF; }   // copy block: no transform runs (device convention)
        else if (h.mode & 0x10) h.skipFlags = r.read(8);
        else h.skipFlags = ((h.mode << 4) | 0x0F) & 0xFF;
        const uint32_t dataSize = 1 + ((h.mode >> 5) & 3);
        h.preLen = r.read(8 * dataSize);
        uint64_t maxLen = (uint64_t)a.block_size + a.block_size / 2;   // blockLength + blockLength/2 (:1893)
        if (maxLen < 2048) maxLen = 2048;
        if (maxLen > (1u << 30)) maxLen = 1u << 30;
        if (h.preLen == 0 || h.preLen > maxLen) h.status = KNZ_ERR_BLOCK_SIZE;
        if (a.checksum_bits == 32) h.ck = r.read(32);
        else if (a.checksum_bits == 64) { h.ck = (uint64_t)r.read(32) << 32; h.ck |= r.read(32); }
    }
    return h;
}

// Block headers only (one wave per block): what the host needs before anything is written (lengths, status); the fused
// walk+decode kernel runs after it.
__global__ __launch_bounds__(64) void knz_dec_block_headers_kernel(WalkBlocksArgs a) {
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    KnzWaveReader r;
    r.init(a.stream, a.nbytes, a.blk_bit[b]);
    const KnzBlkHdr h = knz_walk_block_header(a, r);
    int32_t status = h.status;
    const uint32_t chunkSize = (a.entropy == KNZ_E_ANS1 || a.entropy == KNZ_E_FPAQ) ? (4u << 20) : (uint32_t)KNZ_HUF_CHUNK;
    if (status == 0 && (h.preLen + chunkSize - 1) / chunkSize > a.chunks_per_block) status = KNZ_ERR_BLOCK_SIZE;
    if (a.check_out) {
        if (status == 0 && (h.preLen > a.stream_block_size || (h.preLen > a.out_stride && b + 1 < a.nblocks))) status = KNZ_ERR_PROCESS_BLOCK;   // :1707-1710
        if (status == 0 && (uint64_t)b * a.out_stride + h.preLen > a.out_cap) status = KNZ_ERR_WRITE_FILE;                                    // destination too small
        if (threadIdx.x == 0) a.out_off[b] = a.out_base + (uint64_t)b * a.out_stride;
    }
    if (threadIdx.x == 0) {
        a.blk_pre_len[b] = h.preLen;
        a.blk_mode[b] = (uint8_t)h.mode;
        a.blk_skip[b] = (uint8_t)h.skipFlags;
        a.blk_cksum[b] = h.ck;
        a.blk_status[b] = status;
        a.blk_end_bit[b] = r.tell();
    }
}

// The walk of block b by threads 0..127 of a workgroup (sh.s_sync zeroed and a barrier passed). FUSED: the chunk positions
// are consumed by decoder workgroups of the same launch as soon as they are stored.
template <bool FUSED>
__device__ __forceinline__ void knz_walk_block_body(const WalkBlocksArgs& a, const uint32_t b, KnzWalkShared& sh) {
    uint8_t (&s_lut)[1 << KNZ_EXPG_WIN] = sh.s_lut;
    uint32_t (&s_ring)[KNZ_WALK_RING] = sh.s_ring;
    uint32_t (&s_sync)[4] = sh.s_sync;
    volatile uint32_t* vsync = s_sync;
    const uint64_t ringOrigin = (a.blk_bit[b] >> 5) & ~(uint64_t)255;  // ring word r holds stream word ringOrigin + r (mod ring size)
    const bool ringOk = (((uintptr_t)a.stream) & 15) == 0;             // 16-byte loads; otherwise the serial parser runs
    bool ringLive = ringOk;                                            // cleared when the feeder does not answer in time: serial parser from there on
    if (FUSED) wave_raise_priority();                 // the decoders of the same launch share the CU with this chain
    if (threadIdx.x >= 64) {
        // ---- feeder ---------------------------------------------------------------------------------------------------------
        if (!ringOk || a.entropy != KNZ_E_HUFFMAN) return;             // nothing to feed: the other walks read the stream directly
        const int fl = (int)threadIdx.x - 64;
        const uint32_t* swords = (const uint32_t*)a.stream;
        const uint64_t snwords = (a.nbytes + 3) >> 2;
        uint32_t hi = 0;                                                // relative words filled
        uint32_t idle = 0;
        while (vsync[2] == 0) {
            const uint32_t cons = vsync[1];
            // next granules: wanted (inside the look-ahead), allowed (not over [cons, ..)), inside the stream. Up to 8 loads are
            // issued before the first one is stored: one granule per HBM round trip could not keep up with the walker.
            uint32_t g = 0;
            if (hi < cons + KNZ_WALK_AHEAD && ringOrigin + hi < snwords) {
                const uint32_t room = ((cons & ~255u) + KNZ_WALK_RING - hi) >> 8;          // granules that fit without touching cons
                const uint32_t want = (cons + KNZ_WALK_AHEAD - hi + 255) >> 8;
                const uint32_t left = (uint32_t)min((uint64_t)8, (snwords - ringOrigin - hi + 255) >> 8);
                g = min(min(room, want), left);
            }
            if (g) {
                uint32_t pf[8][4];
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if ((uint32_t)q < g) { const uint4 v = knz_walk_load_granule(swords, snwords, ringOrigin + hi + 256 * q, fl); pf[q][0] = v.x; pf[q][1] = v.y; pf[q][2] = v.z; pf[q][3] = v.w; }
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if ((uint32_t)q < g) { uint4 v; v.x = knz_bswap32(pf[q][0]); v.y = knz_bswap32(pf[q][1]); v.z = knz_bswap32(pf[q][2]); v.w = knz_bswap32(pf[q][3]); *(uint4*)&s_ring[(hi + 256 * q + 4 * (uint32_t)fl) & (KNZ_WALK_RING - 1)] = v; }
                hi += 256 * g;
                wg_fence_release();
                if (fl == 0) vsync[0] = hi;
                idle = 0;
            } else {
                wave_spin_pause();
                if (++idle > (1u << 26)) break;                         // never in practice: the walker sets s_sync[2]
            }
        }
        return;
    }
    for (uint32_t i = threadIdx.x; i < (1u << KNZ_EXPG_WIN); i += 64) s_lut[i] = (uint8_t)knz_expg_lut_entry(i);
    wave_sync_lds();
    const bool writer = threadIdx.x == 0;            // every lane runs the (uniform) parse, lane 0 stores the results
    KnzWaveReader r;
    const uint64_t start = a.blk_bit[b];
    const uint64_t end = start + a.blk_bits[b];
    // the block-local stream is its own bitstream in the reference (r = (read+7)>>3 bytes): reads past `end`
    // rounded up to a byte are an EOS panic there; checked below per chunk
    r.init(a.stream, a.nbytes, start);
    const KnzBlkHdr bh = knz_walk_block_header(a, r);
    int32_t status = bh.status;
    const uint32_t mode = bh.mode, skipFlags = bh.skipFlags, preLen = bh.preLen, entropy = bh.entropy;
    const uint64_t ck = bh.ck;
    uint32_t published = 0;                           // chunk positions stored so far
    if (writer) {
        a.blk_pre_len[b] = preLen;
        a.blk_mode[b] = (uint8_t)mode;
        a.blk_skip[b] = (uint8_t)skipFlags;
        a.blk_cksum[b] = ck;
    }
    const uint64_t limit = start + (((a.blk_bits[b] + 7) >> 3) << 3);
    const uint32_t cpb = a.chunks_per_block;
    if (status == 0) {
        const uint32_t chunkSize = (a.entropy == KNZ_E_ANS1 || a.entropy == KNZ_E_FPAQ) ? (4u << 20) : (uint32_t)KNZ_HUF_CHUNK;
        const uint32_t nchunks = (preLen + chunkSize - 1) / chunkSize;
        if (nchunks > cpb) status = KNZ_ERR_BLOCK_SIZE;
        // Huffman chunks: the stream is pulled through an LDS ring AHEAD of the walk (the loads of the next 26 KiB are issued
        // before a header is parsed and land in the ring after it), so that the chain chunk k -> chunk k+1 never waits for
        // HBM, and the header itself is parsed by the whole wave (knz_huf_parse_header_wave).
        const uint32_t* swords = (const uint32_t*)a.stream;
        const uint64_t snwords = (a.nbytes + 3) >> 2;
        uint64_t pos = r.tell();                                           // walk position, authoritative
        bool stale = false;                                                // r is behind pos
        uint32_t filled = 0;                                               // last feeder progress seen
        for (uint32_t k = 0; k < nchunks && status == 0; k++) {
            const uint32_t sz = min(chunkSize, preLen - k * chunkSize);
            if (writer) knz_publish64(&a.chunk_bit[(size_t)b * cpb + k], pos);
            published = k + 1;
            if (entropy == KNZ_E_HUFFMAN && sz >= 32 && ringLive) {
                KNZ_PROF_T(w0);
                const uint64_t w0i = pos >> 5;
                const uint32_t rel = (uint32_t)(w0i - ringOrigin);
                if (writer) vsync[1] = rel;                             // everything before this word may be overwritten
                // the feeder is normally far ahead (its progress is re-read only when the last value seen does not cover the
                // header window); wait for the window otherwise (bounded: a stuck feeder = error)
                const uint32_t need = (uint32_t)min((uint64_t)rel + KNZ_HW_WORDS, ((snwords - ringOrigin + 255) & ~(uint64_t)255));
                if (filled < need) {
                    uint32_t spins = 0;
                    while ((filled = vsync[0]) < need) { wave_spin_pause(); if (++spins > (1u << 24)) break; }
                    // a feeder that does not answer (emulator, debugger, oversubscribed CU) is a timing event, not a damaged stream:
                    // the walk goes on with the serial parser, which reads the stream directly
                    if (filled < need) ringLive = false;
                    else wg_fence_acquire();
                }
              if (ringLive) {
                KNZ_PROF_T(w1);
                KNZ_PROF_T(w2);
                // the ring holds byte-swapped words (the feeder swaps): the header is parsed in place
                const KnzHufHdr hdr = knz_huf_parse_header_wave<false>(KnzRingWin{s_ring, rel, KNZ_WALK_RING - 1}, (uint32_t)(pos & 31), nullptr, nullptr, (int)threadIdx.x);
                KNZ_PROF_T(w3);
                KNZ_PROF_T(w4);
                KNZ_PROF_ADD(8, w0, w1); KNZ_PROF_ADD(9, w1, w2); KNZ_PROF_ADD(10, w2, w3); KNZ_PROF_ADD(11, w3, w4);
                if (hdr.status == 2) { status = KNZ_ERR_PROCESS_BLOCK; break; }
                if (hdr.status == 0) {
                    uint64_t np = (w0i << 5) + hdr.end;
                    if (hdr.count > 1) {
                        if ((int32_t)(hdr.fb[0] | hdr.fb[1] | hdr.fb[2] | hdr.fb[3]) < 0) status = KNZ_ERR_PROCESS_BLOCK;
                        np += (uint64_t)hdr.fb[0] + hdr.fb[1] + hdr.fb[2] + hdr.fb[3] + 8ull * (sz & 3);
                    }
                    pos = np;
                    stale = true;
                    if (pos > limit) status = KNZ_ERR_PROCESS_BLOCK;
                    continue;
                }
                // unusual header: serial parser below
              }
            }
            if (stale) { r.seek(pos); stale = false; }
            if (entropy == KNZ_E_NONE || (entropy == KNZ_E_HUFFMAN && sz < 32) || ((entropy == KNZ_E_ANS0 || entropy == KNZ_E_ANS1) && preLen <= 32)) {
                r.seek(r.tell() + 8ull * sz);                          // raw bytes (HuffmanCodec.go:769-771, ANSRangeCodec.go:720-723)
            } else if (entropy == KNZ_E_FPAQ) {                       // FPAQDecoder.Read :357-377
                const uint32_t szb = knz_read_varint(r);
                if ((int32_t)szb < 0 || (uint64_t)szb >= 2ull * preLen) status = KNZ_ERR_PROCESS_BLOCK;
                r.seek(r.tell() + 56 + 8ull * szb);
            } else if (entropy == KNZ_E_ANS1) {
                uint32_t lr; int total;
                if (!knz_ans1_parse_header(r, nullptr, lr, total, a.ans1_ctx_bit ? a.ans1_ctx_bit + ((size_t)b * cpb + k) * 257 : nullptr, writer) || total == 0) { status = KNZ_ERR_PROCESS_BLOCK; break; }
                const uint32_t szb = knz_read_varint(r);
                if (szb >= (1u << 27)) status = KNZ_ERR_PROCESS_BLOCK;
                r.seek(r.tell() + 128 + 8ull * szb);
            } else if (entropy == KNZ_E_ANS0) {
                // decodeHeader (ANSRangeCodec.go:605-710) far enough to find the end of the chunk
                const uint32_t lr = 8 + r.read(3);
                uint32_t llr = 3;
                while ((1u << llr) <= lr) llr++;
                uint32_t count;
                if (r.read(1) == 0) count = r.read(1) == 1 ? 0 : 256;
                else {
                    uint32_t lastMask = r.read(5);
                    count = 0;
                    for (uint32_t m = 0; m <= lastMask; m++) count += (uint32_t)__popc(r.read(8));
                }
                if (count == 0) { status = KNZ_ERR_PROCESS_BLOCK; break; }
                const uint32_t chk = count < 64 ? 6 : 8;
                for (uint32_t i = 1; i < count; i += chk) {
                    const uint32_t logMax = r.read(llr);
                    const uint32_t endj = min(i + chk, count);
                    if (logMax > 16) { status = KNZ_ERR_PROCESS_BLOCK; break; }
                    r.seek(r.tell() + (uint64_t)(endj - i) * logMax);
                }
                if (count > 1 && status == 0) {
                    const uint32_t szb = knz_read_varint(r);
                    if (szb >= (1u << 27)) status = KNZ_ERR_PROCESS_BLOCK;
                    r.seek(r.tell() + 128 + 8ull * szb);
                }
            } else {
                // alphabet (EntropyUtils.go:71-119)
                uint32_t count;
                if (r.read(1) == 0) {
                    if (r.read(1) == 1) { status = KNZ_ERR_PROCESS_BLOCK; break; }  // empty alphabet: Read returns short
                    count = 256;
                } else {
                    uint32_t lastMask = r.read(5);
                    count = 0;
                    for (uint32_t bitsLeft = 8 * (lastMask + 1); bitsLeft > 0;) {
                        const uint32_t take = bitsLeft > 32 ? 32 : bitsLeft;
                        count += (uint32_t)__popc(r.read(take));
                        bitsLeft -= take;
                    }
                    if (count == 0) { status = KNZ_ERR_PROCESS_BLOCK; break; }
                }
                for (uint32_t i = 0; i < count;) {
                    const uint32_t e = s_lut[r.peek(KNZ_EXPG_WIN)];
                    const uint32_t nc = e >> 4;
                    if (nc == 0 || i + nc > count) { knz_skip_expg(r); i++; }
                    else { r.skip(e & 15); i += nc; }
                }
                if (count > 1) {
                    uint64_t fb = 0;
                    for (int j = 0; j < 4; j++) {
                        uint32_t v = knz_read_varint(r);
                        if ((int32_t)v < 0) status = KNZ_ERR_PROCESS_BLOCK;
                        fb += v;
                    }
                    r.seek(r.tell() + fb + 8ull * (sz & 3));
                }
            }
            pos = r.tell();
            if (pos > limit) status = KNZ_ERR_PROCESS_BLOCK;            // ran past the block payload
        }
        if (writer) a.blk_end_bit[b] = pos;
    } else if (writer) a.blk_end_bit[b] = r.tell();
    if (FUSED) {
        // a failed walk never reaches the remaining chunks: release their decoders. The status was preset by the header pass
        // and the decoders may have put an error there already: only an error is stored.
        if (status != 0) {
            for (uint32_t k = published + threadIdx.x; k < cpb; k += 64) knz_publish64(&a.chunk_bit[(size_t)b * cpb + k], KNZ_CHUNK_ERR);
            if (writer) a.blk_status[b] = status;
        }
        if (writer) vsync[2] = 1;
    } else if (writer) { a.blk_status[b] = status; vsync[2] = 1; }
}

__global__ __launch_bounds__(128) void knz_dec_walk_blocks_kernel(WalkBlocksArgs a) {
    __shared__ KnzWalkShared sh;
    const uint32_t b = blockIdx.x;
    if (b >= a.nblocks) return;
    if (threadIdx.x < 4) sh.s_sync[threadIdx.x] = 0;
    __syncthreads();
    knz_walk_block_body<false>(a, b, sh);
}

// Walk and decode in ONE launch: workgroups [0, nblocks) are the walkers (dispatched first, 2 of their 4 waves leave at once),
// workgroup nblocks + k * nblocks + b decodes chunk k of block b as soon as walker b has stored its position (chunk-major
// order: the walkers of all blocks advance together at ~4 us per chunk, so the decoders become runnable in dispatch order).
// chunk_bit is preset to KNZ_CHUNK_NOT_READY by the host. A decoder that gives up waiting hands its chunk to the serial
// kernel that follows in the stream (never seen: the walkers are resident before the first decoder is dispatched).
__global__ __launch_bounds__(256) void knz_huf_walk_decode_kernel(WalkBlocksArgs wa, HufDecArgs da, uint8_t* fallback) {
    __shared__ union KnzWalkDecodeShared { KnzHufParShared d; KnzWalkShared w; } sh;
    __shared__ uint64_t s_cbit;
    const uint32_t nblocks = wa.nblocks;
    if (blockIdx.x < nblocks) {
        if (threadIdx.x < 4) sh.w.s_sync[threadIdx.x] = 0;
        __syncthreads();
        if (threadIdx.x >= 128) return;
        knz_walk_block_body<true>(wa, blockIdx.x, sh.w);
        return;
    }
    const uint32_t id = blockIdx.x - nblocks, cpb = da.chunks_per_block;
    const uint32_t k = id / nblocks, b = id % nblocks;
    const uint32_t slot = b * cpb + k;
    if ((uint64_t)k * KNZ_HUF_CHUNK >= da.blk_pre_len[b] || da.blk_status[b] != 0) {   // lengths and status: header pass
        if (threadIdx.x == 0) fallback[slot] = 0;
        return;
    }
    if (threadIdx.x == 0) {
        uint64_t v = knz_poll64(&da.chunk_bit[slot]);
        for (uint32_t spins = 0; v == KNZ_CHUNK_NOT_READY && spins < (1u << 21); spins++) { wg_spin_pause(); v = knz_poll64(&da.chunk_bit[slot]); }
        s_cbit = v;
    }
    __syncthreads();
    const uint64_t cbit = s_cbit;
    if (cbit == KNZ_CHUNK_ERR) { if (threadIdx.x == 0) fallback[slot] = 0; return; }
    if (cbit == KNZ_CHUNK_NOT_READY) { if (threadIdx.x == 0) fallback[slot] = 1; return; }
    knz_huf_decode_par_body(da, fallback, slot, cbit, sh.d);
}

