// SRT (sorted rank transform) and LZP (Lempel-Ziv predict) of kanzi bitstream v6 on gfx950.
// Replaces SRT.Forward / Inverse / preprocess / encodeHeader / decodeHeader (v2/transform/SRT.go:49-312) and
// LZPCodec.Forward / Inverse / findMatch (v2/transform/LZCodec.go:982-1207).
//
// Both are one dependent chain per block by definition (a move-to-front list, a one-entry-per-hash prediction table that every
// position rewrites), so a block is one wave; the wave's 64 lanes are used where the chain allows it:
//  * SRT forward: histogram / first occurrences / bucket layout in parallel, then one list access per RUN (the run heads of a
//    4 KiB tile are found with one ballot per 64 bytes), the list in lane-transposed registers (SbrtWave<1>), the bucket
//    cursors in registers too; the zeros behind a run head are a bulk fill done up front.
//  * SRT inverse: the 256 bucket streams are read through 32-byte LDS windows, a run of zeros in the current bucket is found
//    with one ballot and written by the whole wave; a non-zero rank rotates the register list with one DPP shift.
//  * LZP forward: 64 positions per step are hashed, looked up and tested for a 64-byte prediction at once under the assumption
//    that all of them are literals (true for all but a few windows: a prediction has to hold for 64 bytes to count); positions
//    of the same window that share a hash are resolved among the lanes. Only the first candidate of a window is examined
//    serially (wave-wide match length).
//  * LZP inverse: a literal needs no table look-up unless it is the 0xFC flag byte, so runs of non-flag bytes are copied and
//    hashed 64 at a time; flags (escapes and matches) are handled one by one, matches are wave-wide (periodic) copies.
#include "bits.h"

#define KNZ_LZP_HASH_LOG 16
#define KNZ_LZP_SEED 0x7FEB352Du
#define KNZ_LZP_MIN_MATCH 64
#define KNZ_LZP_FLAG 0xFCu
#define KNZ_LZP_MIN_BLOCK 128
#define KNZ_SRT_HEADER_MAX (4 * 256)

// ---------------------------------------------------------------------------------------------------------------------
// 256 counters, one per symbol, lane-transposed: symbol c lives in register c >> 6 of lane c & 63 (indices are wave-uniform)
struct KnzSymRegs {
    uint32_t v[4];
    __device__ __forceinline__ uint32_t get(uint32_t c) const {
        const uint32_t l = c & 63;
        switch (c >> 6) {
            case 0: return wave_readlane(v[0], l);
            case 1: return wave_readlane(v[1], l);
            case 2: return wave_readlane(v[2], l);
            default: return wave_readlane(v[3], l);
        }
    }
    __device__ __forceinline__ void add(uint32_t c, uint32_t d, int lane) {
        const bool me = (uint32_t)lane == (c & 63);
        switch (c >> 6) {
            case 0: v[0] += me ? d : 0u; break;
            case 1: v[1] += me ? d : 0u; break;
            case 2: v[2] += me ? d : 0u; break;
            default: v[3] += me ? d : 0u; break;
        }
    }
};

// bucket layout shared by both directions: symbols by decreasing frequency, ties by increasing symbol (SRT.go:134-167 is a shell
// sort with exactly this order). Lane l handles symbols l, l+64, l+128, l+192. start = first index of the symbol's bucket.
__device__ __forceinline__ void knz_srt_layout(const int* s_freq, int lane, KnzSymRegs& start, uint32_t& nbSymbols, uint8_t* s_sorted) {
    uint32_t present = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = 64 * k + lane;
        const int f = s_freq[c];
        uint32_t before = 0, rank = 0;
        for (int d = 0; d < 256; d++) {
            const int fd = s_freq[d];
            if (fd > 0 && (fd > f || (fd == f && d < c))) { before += (uint32_t)fd; rank++; }
        }
        start.v[k] = before;
        if (f > 0) { present++; if (s_sorted) s_sorted[rank] = (uint8_t)c; }
    }
    nbSymbols = wave_reduce_add(present);
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void knz_srt_forward_kernel(XfArgs a) {
    __shared__ int s_freq[256];
    __shared__ uint32_t s_first[256];
    __shared__ uint8_t s_r2s[256];
    __shared__ uint32_t s_hoff[5];
    __shared__ uint8_t s_tile[4096 + 4];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    if (n == 0) { if (lane == 0) { a.ok[b] = 1; a.out_len[b] = 0; } return; }
    if ((uint64_t)a.out_cap < (uint64_t)n + KNZ_SRT_HEADER_MAX) { if (lane == 0) { a.ok[b] = 0; a.out_len[b] = 0; } return; }   // :58-60
#pragma unroll
    for (int k = 0; k < 4; k++) { s_freq[64 * k + lane] = 0; s_first[64 * k + lane] = 0xFFFFFFFFu; s_r2s[64 * k + lane] = 0; }
    wave_sync();
    // occurrences and first positions (:66-82)
    for (uint32_t i = lane; i < n; i += 64) { const uint32_t c = src[i]; atomicAdd(&s_freq[c], 1); atomicMin(&s_first[c], i); }
    wave_sync();
    // initial list = symbols in order of first appearance
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = 64 * k + lane;
        if (s_freq[c] > 0) {
            const uint32_t fp = s_first[c];
            uint32_t ra = 0;
            for (int d = 0; d < 256; d++) ra += (s_freq[d] > 0 && s_first[d] < fp) ? 1u : 0u;
            s_r2s[ra] = (uint8_t)c;
        }
    }
    KnzSymRegs bk;
    uint32_t nbSymbols;
    knz_srt_layout(s_freq, lane, bk, nbSymbols, nullptr);
    // header: 256 varints (:261-275); symbol order = index order, register k of lane l is symbol 64k + l
    uint32_t hs = 0;
    {
        uint32_t hl[4], ho[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t f = (uint32_t)s_freq[64 * k + lane], l = 1;
            while (f >= 128) { l++; f >>= 7; }
            hl[k] = l;
        }
        uint32_t basek = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t incl = wave_scan_incl(hl[k]);
            ho[k] = basek + incl - hl[k];
            basek += wave_bcast(incl, 63);
        }
        hs = basek;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t f = (uint32_t)s_freq[64 * k + lane], o = ho[k];
            while (f >= 128) { dst[o++] = (uint8_t)(0x80 | (f & 0x7F)); f >>= 7; }
            dst[o] = (uint8_t)f;
        }
    }
    uint8_t* out = dst + hs;
    // every byte that is not a run head is rank 0 (:118-122)
    for (uint32_t i = lane; i < n; i += 64) out[i] = 0;
    wave_sync();
    SbrtWave<1> w;
#pragma unroll
    for (int k = 0; k < 4; k++) { w.s[k] = s_r2s[64 * k + lane]; w.q[k] = 0; w.p[k] = 0; }
    uint32_t prevC = 0, prevHead = 0;
    bool havePrev = false;
    int t = 0;
    for (uint32_t base = 0; base < n; base += 4096) {
        const uint32_t cnt = min(4096u, n - base);
        wave_sync();
        if (lane == 0) s_tile[0] = base ? src[base - 1] : (uint8_t)0;
        for (uint32_t i = lane; i < cnt; i += 64) s_tile[1 + i] = src[base + i];
        wave_sync();
        for (uint32_t g = 0; g < cnt; g += 64) {
            const bool valid = g + lane < cnt;
            const uint32_t cur = valid ? s_tile[1 + g + lane] : 0u, prv = valid ? s_tile[g + lane] : 0u;
            const bool head = valid && (base + g + lane == 0 || cur != prv);
            uint64_t m = wave_ballot(head);
            while (m) {
                const uint32_t l = (uint32_t)(__ffsll((unsigned long long)m) - 1);
                m &= m - 1;
                const uint32_t c = wave_readlane(cur, l);
                const uint32_t pos = base + g + l;
                if (havePrev) bk.add(prevC, pos - prevHead, lane);
                const uint32_t r = w.template step_any<true>(c, ++t, lane);
                const uint32_t p = bk.get(c);
                if (lane == 0 && r) out[p] = (uint8_t)r;
                prevC = c; prevHead = pos; havePrev = true;
            }
        }
    }
    if (lane == 0) { a.ok[b] = 1; a.out_len[b] = n + hs; }
}

// ---------------------------------------------------------------------------------------------------------------------
// SRT forward without the chain. The rank SRT writes for a run head is the move-to-front rank of that position, and a
// move-to-front rank does not depend on how the list started except at a symbol's first occurrence (seen symbols are always in
// front of unseen ones): so the ranks are the MTFT ranks of the segment-parallel knz_sbrt_* kernels (mode 1), with the first
// occurrence of the k-th new symbol replaced by k, and every byte that is not a run head has MTFT rank 0 anyway. What is left is
// SRT's layout: a stable partition of the positions by symbol (bucket of c = its ranks in order), done per 8 KiB segment with
// per-segment symbol counts, a scan over the segments and a match-any inside each row of 64 positions.
struct SrtParArgs {
    uint32_t nblocks, segs_per_block;
    const uint64_t* in_ptr; const uint32_t* in_len; const uint64_t* out_ptr; uint32_t out_cap;
    uint32_t* out_len; int32_t* ok; const uint8_t* active;
    uint32_t* tab;                 // [nblocks * KNZ_SRT_TAB]: start[256], firstPos[256], firstRank[256], hs
    const uint64_t* rank_ptr;      // [nblocks] MTFT ranks of every position (knz_sbrt_apply_kernel<1>)
    int32_t* seg_cnt;              // [nblocks * segs_per_block * 256]
};
#define KNZ_SRT_TAB (3 * 256 + 4)

__global__ __launch_bounds__(256) void knz_srt_stats_kernel(SrtParArgs a) {
    __shared__ int s_freq[256];
    __shared__ uint32_t s_first[256];
    __shared__ uint32_t s_hoff[257];
    const int c = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    if (n == 0) { if (c == 0) { a.ok[b] = 1; a.out_len[b] = 0; } return; }
    if ((uint64_t)a.out_cap < (uint64_t)n + KNZ_SRT_HEADER_MAX) { if (c == 0) { a.ok[b] = 0; a.out_len[b] = 0; } return; }   // SRT.go:58-60
    s_freq[c] = 0; s_first[c] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t i = c; i < n; i += 256) { const uint32_t v = src[i]; atomicAdd(&s_freq[v], 1); atomicMin(&s_first[v], i); }
    __syncthreads();
    const int f = s_freq[c];
    const uint32_t fp = s_first[c];
    uint32_t before = 0, ra = 0;
    for (int d = 0; d < 256; d++) {
        const int fd = s_freq[d];
        if (fd > 0 && (fd > f || (fd == f && d < c))) before += (uint32_t)fd;      // buckets by decreasing frequency, ties by symbol (:134-167)
        if (fd > 0 && s_first[d] < fp) ra++;
    }
    uint32_t hl = 1;
    for (uint32_t v = (uint32_t)f; v >= 128; v >>= 7) hl++;
    s_hoff[c + 1] = hl;
    __syncthreads();
    if (c == 0) { s_hoff[0] = 0; for (int d = 1; d <= 256; d++) s_hoff[d] += s_hoff[d - 1]; }
    __syncthreads();
    {
        uint32_t v = (uint32_t)f, o = s_hoff[c];                                  // encodeHeader :261-275
        while (v >= 128) { dst[o++] = (uint8_t)(0x80 | (v & 0x7F)); v >>= 7; }
        dst[o] = (uint8_t)v;
    }
    uint32_t* t = a.tab + (size_t)b * KNZ_SRT_TAB;
    t[c] = before; t[256 + c] = fp; t[512 + c] = ra;
    if (c == 0) { t[768] = s_hoff[256]; a.ok[b] = 1; a.out_len[b] = n + s_hoff[256]; }
}

__global__ __launch_bounds__(64) void knz_srt_seg_count_kernel(SrtParArgs a) {
    __shared__ int s_cnt[256];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    int32_t* out = a.seg_cnt + (size_t)blockIdx.x * 256;
    for (int i = lane; i < 256; i += 64) s_cnt[i] = 0;
    wave_sync();
    if (lo < n) {
        const uint32_t hi = min(n, lo + KNZ_SEG);
        const uint8_t* src = (const uint8_t*)a.in_ptr[b];
        for (uint32_t i = lo + lane; i < hi; i += 64) atomicAdd(&s_cnt[src[i]], 1);
    }
    wave_sync();
    for (int i = lane; i < 256; i += 64) out[i] = s_cnt[i];
}

__global__ __launch_bounds__(256) void knz_srt_seg_scan_kernel(SrtParArgs a) {
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const int c = threadIdx.x;
    const uint32_t nseg = (a.in_len[b] + KNZ_SEG - 1) / KNZ_SEG;
    int run = 0;
    for (uint32_t s = 0; s < nseg; s++) {
        int32_t* p = a.seg_cnt + ((size_t)b * a.segs_per_block + s) * 256 + c;
        const int v = *p;
        *p = run;
        run += v;
    }
}

__global__ __launch_bounds__(64) void knz_srt_scatter_kernel(SrtParArgs a) {
    __shared__ uint32_t s_pos[256];                                               // next free entry of every bucket, for this segment
    __shared__ uint32_t s_firstPos[256];
    __shared__ uint8_t s_firstRank[256];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b] || a.ok[b] != 1) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG);
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    const uint8_t* rk = (const uint8_t*)a.rank_ptr[b];
    const uint32_t* t = a.tab + (size_t)b * KNZ_SRT_TAB;
    uint8_t* out = (uint8_t*)a.out_ptr[b] + t[768];
    const int32_t* pre = a.seg_cnt + (size_t)blockIdx.x * 256;
    for (int i = lane; i < 256; i += 64) { s_pos[i] = t[i] + (uint32_t)pre[i]; s_firstPos[i] = t[256 + i]; s_firstRank[i] = (uint8_t)t[512 + i]; }
    wave_sync();
    for (uint32_t i0 = lo; i0 < hi; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        const bool valid = i < hi;
        const uint32_t c = valid ? src[i] : 0u;
        uint32_t r = valid ? rk[i] : 0u;
        if (valid && i == s_firstPos[c]) r = s_firstRank[c];                      // the k-th new symbol has rank k (:66-75)
        // lanes of this row with my symbol
        uint64_t peers = wave_ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; bit++) {
            const uint64_t bal = wave_ballot(((c >> bit) & 1) != 0);
            peers &= ((c >> bit) & 1) ? bal : ~bal;
        }
        const uint32_t base = valid ? s_pos[c] : 0u;
        const uint32_t before = (uint32_t)__popcll(peers & (((uint64_t)1 << lane) - 1));
        if (valid) out[base + before] = (uint8_t)r;
        wave_sync_lds();                                                          // every lane has read s_pos before it moves
        if (valid && (peers >> lane) == 1) s_pos[c] = base + (uint32_t)__popcll(peers);   // the last lane of the group
        wave_sync_lds();
    }
}

__global__ void knz_fill_ptrs_kernel(uint32_t nblocks, uint64_t base, uint64_t stride, uint64_t* ptrs) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblocks) ptrs[b] = base + (uint64_t)b * stride;
}

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void knz_srt_inverse_kernel(XfArgs a) {
    __shared__ int s_freq[256];
    __shared__ uint8_t s_hdr[KNZ_SRT_HEADER_MAX];
    __shared__ uint8_t s_sorted[256];
    __shared__ uint8_t s_fb[256];
    __shared__ uint8_t s_r2s[256];
    __shared__ uint8_t s_win[256 * 32];
    __shared__ uint32_t s_hs;
    __shared__ int s_err;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    if (n == 0) { if (lane == 0) { a.ok[b] = 1; a.out_len[b] = 0; } return; }
    for (uint32_t i = lane; i < KNZ_SRT_HEADER_MAX; i += 64) s_hdr[i] = i < n ? src[i] : (uint8_t)0;
#pragma unroll
    for (int k = 0; k < 4; k++) s_r2s[64 * k + lane] = 0;
    wave_sync();
    if (lane == 0) {                                                   // decodeHeader :277-312 (reads past the block = Go panic)
        uint32_t h = 0;
        int err = 0;
        for (int i = 0; i < 256; i++) {
            if (h >= n) { err = 1; break; }
            int val = s_hdr[h++];
            if (val < 128) { s_freq[i] = val; continue; }
            int res = val & 0x7F;
            if (h >= n) { err = 1; break; }
            val = s_hdr[h++]; res |= (val & 0x7F) << 7;
            if (val >= 128) {
                if (h >= n) { err = 1; break; }
                val = s_hdr[h++]; res |= (val & 0x7F) << 14;
                if (val >= 128) { if (h >= n) { err = 1; break; } val = s_hdr[h++]; res |= (val & 0x7F) << 21; }
            }
            s_freq[i] = res;
        }
        s_hs = h; s_err = err;
    }
    wave_sync();
    const uint32_t hs = s_hs;
    bool bad = s_err != 0;
    const uint32_t len = bad ? 0u : n - hs;
    if (!bad && len > a.out_cap) bad = true;                           // :187-189
    uint32_t total = 0;
    if (!bad) {
#pragma unroll
        for (int k = 0; k < 4; k++) total += (uint32_t)s_freq[64 * k + lane];
        total = wave_reduce_add(total);
        // a block whose frequencies do not add up to its length is damaged (the reference walks its output buffer to the end
        // with whatever the buckets hold; the device reports the block instead, docs/HISTORY.md section 2)
        if (total != len) bad = true;
    }
    if (bad) { if (lane == 0) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; a.out_len[b] = 0; } return; }
    if (len == 0) { if (lane == 0) { a.ok[b] = 1; a.out_len[b] = 0; } return; }
    const uint8_t* in = src + hs;
    KnzSymRegs bk, be;
    uint32_t nbSymbols;
    knz_srt_layout(s_freq, lane, bk, nbSymbols, s_sorted);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int c = 64 * k + lane;
        be.v[k] = bk.v[k] + (uint32_t)s_freq[c];
        if (s_freq[c] > 0) s_fb[c] = in[bk.v[k]];                      // the bucket's first entry: the symbol's initial rank
    }
    wave_sync();
    if (lane == 0) for (uint32_t i = 0; i < nbSymbols; i++) { const uint32_t c = s_sorted[i]; s_r2s[s_fb[c]] = (uint8_t)c; }   // :200-210, in that order
    wave_sync();
#pragma unroll
    for (int k = 0; k < 4; k++) if (s_freq[64 * k + lane] > 0) bk.v[k] += 1;
    // bucket windows: 32 bytes of every bucket from its read position
    uint32_t wb[4];                                                    // window base per symbol (same register layout)
#pragma unroll
    for (int k = 0; k < 4; k++) wb[k] = bk.v[k];
    for (uint32_t c = 0; c < 256; c += 2) {
        const uint32_t cc = c + ((uint32_t)lane >> 5), o = (uint32_t)lane & 31;
        KnzSymRegs t1; t1.v[0] = bk.v[0]; t1.v[1] = bk.v[1]; t1.v[2] = bk.v[2]; t1.v[3] = bk.v[3];
        const uint32_t p0 = t1.get(c), p1 = t1.get(c + 1);
        t1.v[0] = be.v[0]; t1.v[1] = be.v[1]; t1.v[2] = be.v[2]; t1.v[3] = be.v[3];
        const uint32_t e0 = t1.get(c), e1 = t1.get(c + 1);
        const uint32_t p = (lane >> 5) ? p1 : p0, e = (lane >> 5) ? e1 : e0;
        s_win[cc * 32 + o] = p + o < e ? in[p + o] : (uint8_t)0;
    }
    wave_sync();
    uint32_t r2s[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r2s[k] = s_r2s[64 * k + lane];
    KnzSymRegs wbr; wbr.v[0] = wb[0]; wbr.v[1] = wb[1]; wbr.v[2] = wb[2]; wbr.v[3] = wb[3];
    uint32_t c = wave_bcast(r2s[0], 0);
    uint32_t i = 0;
    while (i < len) {
        const uint32_t pos = bk.get(c), end = be.get(c);
        if (pos >= end) {                                              // bucket exhausted: one more copy, the symbol leaves the list
            if (lane == 0) dst[i] = (uint8_t)c;
            i++;
            if (nbSymbols == 1) { for (uint32_t j = i + lane; j < len; j += 64) dst[j] = (uint8_t)c; i = len; break; }
            nbSymbols--;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t sh = wave_shl1(r2s[k]);
                const uint32_t nxt0 = wave_bcast(r2s[k < 3 ? k + 1 : 3], 0);
                if (lane == 63) sh = k < 3 ? nxt0 : r2s[3];
                if ((uint32_t)(64 * k + lane) < nbSymbols) r2s[k] = sh;
            }
            c = wave_bcast(r2s[0], 0);
            continue;
        }
        uint32_t base = wbr.get(c);
        if (pos - base >= 32) {                                        // refill the window of this bucket at its read position
            wave_sync_lds();
            if (lane < 32) s_win[c * 32 + lane] = pos + lane < end ? in[pos + lane] : (uint8_t)0;
            wbr.add(c, pos - base, lane);
            base = pos;
            wave_sync();
        }
        const uint32_t off = pos - base;
        const uint32_t avail = min(32u - off, end - pos);
        const uint32_t bl = (uint32_t)lane < avail ? s_win[c * 32 + off + lane] : 0u;
        const uint64_t nz = wave_ballot((uint32_t)lane < avail && bl != 0);
        const uint32_t z = nz ? (uint32_t)(__ffsll((unsigned long long)nz) - 1) : avail;
        if (nz == 0) {                                                 // zeros only: the symbol repeats, keep reading its bucket
            const uint32_t cp = min(z, len - i);
            if ((uint32_t)lane < cp) dst[i + lane] = (uint8_t)c;
            i += cp;
            bk.add(c, z, lane);
            continue;
        }
        const uint32_t r = wave_readlane(bl, z);
        const uint32_t cp = min(z + 1, len - i);
        if ((uint32_t)lane < cp) dst[i + lane] = (uint8_t)c;
        i += cp;
        bk.add(c, z + 1, lane);
        // ranks 1..r move up by one, the symbol goes to rank r (:229-243)
        if (r < 64) {
            const uint32_t sh = wave_shl1(r2s[0]);
            r2s[0] = (uint32_t)lane < r ? sh : ((uint32_t)lane == r ? c : r2s[0]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                uint32_t sh = wave_shl1(r2s[k]);
                const uint32_t nxt0 = wave_bcast(r2s[k < 3 ? k + 1 : 3], 0);
                if (lane == 63) sh = k < 3 ? nxt0 : r2s[3];
                const uint32_t x = (uint32_t)(64 * k + lane);
                r2s[k] = x < r ? sh : (x == r ? c : r2s[k]);
            }
        }
        c = wave_bcast(r2s[0], 0);
    }
    if (lane == 0) { a.ok[b] = 1; a.out_len[b] = len; }
}

// ---------------------------------------------------------------------------------------------------------------------
// LZP
__device__ __forceinline__ uint32_t knz_lzp_hash(uint32_t ctx) { return (KNZ_LZP_SEED * ctx) >> (32 - KNZ_LZP_HASH_LOG); }

// ctx seen by lane l of a window of literals that starts at position p0 with context `ctx`: the bytes of the window enter one
// by one (ctx = ctx << 8 | byte, LZCodec.go:1029), so from the fifth lane on it is the four bytes in front of the lane
__device__ __forceinline__ uint32_t knz_lzp_lane_ctx(const uint8_t* bytes, uint64_t p0, uint32_t ctx, int lane) {
    if (lane >= 4) {
        const uint8_t* q = bytes + p0 + lane;
        return ((uint32_t)q[-4] << 24) | ((uint32_t)q[-3] << 16) | ((uint32_t)q[-2] << 8) | (uint32_t)q[-1];
    }
    uint32_t c = ctx;
    for (int j = 0; j < lane; j++) c = (c << 8) | bytes[p0 + j];
    return c;
}

// number of equal bytes at src[x..] and src[ref..], 8 bytes at a time up to maxMatch (findMatch :1192-1207), by the whole wave
__device__ __forceinline__ int knz_lzp_find_match(const uint8_t* src, int x, int ref, int maxMatch, int lane) {
    const int nwords = maxMatch >> 3;
    for (int w0 = 0; w0 < nwords; w0 += 64) {
        const int w = w0 + lane;
        uint64_t diff = 0;
        if (w < nwords) diff = knz_le64(src + x + 8 * w) ^ knz_le64(src + ref + 8 * w);
        const uint64_t m = wave_ballot(diff != 0);
        if (m) {
            const uint32_t l = (uint32_t)(__ffsll((unsigned long long)m) - 1);
            const uint32_t lo = wave_readlane((uint32_t)diff, l), hi = wave_readlane((uint32_t)(diff >> 32), l);
            const uint64_t d = ((uint64_t)hi << 32) | lo;
            return 8 * (w0 + (int)l) + (int)((__ffsll((unsigned long long)d) - 1) >> 3);
        }
    }
    return 8 * nwords;
}

// bytes [p0 - 8, p0 + 136) of the block staged in LDS by the wave (one coalesced load per lane and 64 bytes): the window's own
// bytes, the 4 context bytes in front of every lane and the 8 bytes at +56 that a prediction has to match first
#define KNZ_LZP_WIN_BACK 8
#define KNZ_LZP_WIN_BYTES 144
__device__ __forceinline__ void knz_lzp_stage(const uint8_t* src, int count, int p0, uint8_t* s_w, int lane) {
    wave_sync_lds();
    for (int i = lane; i < KNZ_LZP_WIN_BYTES; i += 64) { const int g = p0 - KNZ_LZP_WIN_BACK + i; s_w[i] = (g >= 0 && g < count) ? src[g] : (uint8_t)0; }
    wave_sync();
}
__device__ __forceinline__ uint64_t knz_lds_le64(const uint8_t* p) {
    uint64_t v = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) v |= (uint64_t)p[j] << (8 * j);
    return v;
}

__global__ __launch_bounds__(64) void knz_lzp_forward_kernel(LzArgs a) {
    __shared__ uint8_t s_w[KNZ_LZP_WIN_BYTES];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const int count = (int)a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const uint32_t maxEnc = count <= 1024 ? (uint32_t)count + 16 : (uint32_t)count + (uint32_t)count / 64;
    if (a.out_cap < maxEnc || count < KNZ_LZP_MIN_BLOCK) { if (lane == 0) { a.ok[b] = 0; a.out_len[b] = 0; } return; }   // :989-996
    int32_t* hashes = a.hashes + ((size_t)b << KNZ_LZP_HASH_LOG);
    const int srcEnd = count, dstEnd = count - (count >> 6), mainEnd = srcEnd - KNZ_LZP_MIN_MATCH;
    if (lane < 4) dst[lane] = src[lane];
    uint32_t ctx = knz_le32(src);
    int srcIdx = 4, dstIdx = 4;
    bool skip = false;
    while (srcIdx < srcEnd) {
        if (dstIdx >= dstEnd) { skip = true; break; }
        const bool inMain = srcIdx < mainEnd;
        const int navail = min(64, (inMain ? mainEnd : srcEnd) - srcIdx);
        const bool valid = lane < navail;
        const int q = srcIdx + lane;
        knz_lzp_stage(src, count, srcIdx, s_w, lane);
        const uint8_t* wq = s_w + KNZ_LZP_WIN_BACK + lane;              // my position inside the staged bytes
        // context of my position if everything in front of me in this window is a literal (:1029): the four bytes in front of
        // me from the fifth lane on, the running context shifted by my predecessors' bytes before that
        uint32_t cl;
        if (lane >= 4) cl = ((uint32_t)wq[-4] << 24) | ((uint32_t)wq[-3] << 16) | ((uint32_t)wq[-2] << 8) | (uint32_t)wq[-1];
        else { cl = ctx; for (int j = 0; j < lane; j++) cl = (cl << 8) | s_w[KNZ_LZP_WIN_BACK + j]; }
        const uint32_t h = knz_lzp_hash(cl);
        const int tref = valid ? hashes[h] : 0;
        const uint32_t byte = valid ? (uint32_t)wq[0] : 0u;
        // positions of this window with my hash: the latest one in front of me is my prediction (:1017-1018)
        int dup = -1;
        for (int j = 0; j + 1 < navail; j++) { const uint32_t hj = wave_readlane(h, (uint32_t)j); if (valid && lane > j && h == hj) dup = j; }
        const int ref = dup >= 0 ? srcIdx + dup : tref;
        bool cand = false;
        if (inMain && valid && ref != 0) cand = knz_lds_le64(wq + KNZ_LZP_MIN_MATCH - 8) == knz_le64(src + ref + KNZ_LZP_MIN_MATCH - 8);
        // the reference then measures the prediction and treats anything under 64 bytes as a literal (:1022-1041): settle that
        // here, per lane, so that only a prediction that really holds ends the window (tables of fixed-width records match at
        // +56 again and again without matching in front of it, and a window that ends is a serial step)
        if (cand) {
            uint64_t diff = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) diff |= knz_lds_le64(wq + 8 * k) ^ knz_le64(src + ref + 8 * k);
            cand = diff == 0;
        }
        const uint64_t cm = wave_ballot(cand);
        const int nlit = cm ? (int)(__ffsll((unsigned long long)cm) - 1) : navail;
        const bool lit = lane < nlit;
        const bool esc = lit && ref != 0 && byte == KNZ_LZP_FLAG;       // :1034-1037
        const uint32_t wdt = lit ? (esc ? 2u : 1u) : 0u;
        const uint32_t incl = wave_scan_incl(wdt);
        const uint32_t off = incl - wdt;
        if (wave_ballot(lit && dstIdx + (int)off >= dstEnd) != 0) { skip = true; break; }   // the loop would stop there: no compression
        if (lit) { dst[dstIdx + off] = (uint8_t)byte; if (esc) dst[dstIdx + off + 1] = 0xFF; atomicMax(&hashes[h], q); }
        const uint32_t cnext = (cl << 8) | byte;                        // context after my byte
        if (nlit > 0) ctx = wave_readlane(cnext, (uint32_t)(nlit - 1));
        dstIdx += (int)wave_bcast(incl, 63);
        srcIdx += nlit;
        wave_sync();                                                    // the table entries are in place before the next look-up
        if (nlit < navail) {                                            // a prediction holds here (>= 64 bytes): the reference's loop body, once
            if (dstIdx >= dstEnd) { skip = true; break; }
            const uint32_t h0 = wave_readlane(h, (uint32_t)nlit);
            const int ref0 = (int)wave_readlane((uint32_t)ref, (uint32_t)nlit);
            const uint32_t b0 = wave_readlane(byte, (uint32_t)nlit);
            if (lane == 0) hashes[h0] = srcIdx;
            const int bestLen = knz_lzp_find_match(src, srcIdx, ref0, srcEnd - srcIdx, lane);
            if (bestLen < KNZ_LZP_MIN_MATCH) {
                ctx = (ctx << 8) | b0;
                if (lane == 0) { dst[dstIdx] = (uint8_t)b0; if (ref0 != 0 && b0 == KNZ_LZP_FLAG) dst[dstIdx + 1] = 0xFF; }
                dstIdx += (ref0 != 0 && b0 == KNZ_LZP_FLAG) ? 2 : 1;
                srcIdx++;
            } else {
                srcIdx += bestLen;
                ctx = knz_le32(src + srcIdx - 4);
                int rest = bestLen - KNZ_LZP_MIN_MATCH;
                const int nfe = rest / 254;
                if (dstIdx + 1 + nfe >= dstEnd) { skip = true; break; }   // the 0xFE run reaches dstEnd (:1051-1053): the result is a skip
                if (lane == 0) dst[dstIdx] = (uint8_t)KNZ_LZP_FLAG;
                for (int j = lane; j < nfe; j += 64) dst[dstIdx + 1 + j] = 0xFE;
                if (lane == 0) dst[dstIdx + 1 + nfe] = (uint8_t)(rest - 254 * nfe);
                dstIdx += 2 + nfe;
            }
            wave_sync();
        }
    }
    if (!skip && (srcIdx != count || dstIdx >= dstEnd)) skip = true;    // :1081-1083
    if (lane == 0) { a.ok[b] = skip ? 0 : 1; a.out_len[b] = skip ? 0 : (uint32_t)dstIdx; }
}

__global__ __launch_bounds__(64) void knz_lzp_inverse_kernel(LzArgs a) {
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const long long srcEnd = (long long)a.in_len[b], dstEnd = (long long)a.out_cap;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    if (srcEnd == 0) { if (lane == 0) { a.ok[b] = 1; a.out_len[b] = 0; } return; }
    if (srcEnd < 4 || dstEnd < 4) { if (lane == 0) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; a.out_len[b] = 0; } return; }
    int32_t* hashes = a.hashes + ((size_t)b << KNZ_LZP_HASH_LOG);
    if (lane < 4) dst[lane] = src[lane];
    uint32_t ctx = knz_le32(src);
    long long srcIdx = 4, dstIdx = 4;
    bool bad = false;
    while (srcIdx < srcEnd) {
        const int avail = (int)min((long long)64, srcEnd - srcIdx);
        const bool valid = lane < avail;
        const uint32_t byte = valid ? src[srcIdx + lane] : 0u;
        const uint64_t fm = wave_ballot(valid && byte == KNZ_LZP_FLAG);
        const int nlit = fm ? (int)(__ffsll((unsigned long long)fm) - 1) : avail;
        if (dstIdx + nlit > dstEnd) { bad = true; break; }              // dst[dstIdx] out of range = Go panic
        const uint32_t cl = knz_lzp_lane_ctx(src, (uint64_t)srcIdx, ctx, valid ? lane : 0);   // literals: dst bytes == src bytes
        if (lane < nlit) { dst[dstIdx + lane] = (uint8_t)byte; atomicMax(&hashes[knz_lzp_hash(cl)], (int)(dstIdx + lane)); }
        const uint32_t cnext = (cl << 8) | byte;
        if (nlit > 0) ctx = wave_readlane(cnext, (uint32_t)(nlit - 1));
        srcIdx += nlit; dstIdx += nlit;
        wave_sync();
        if (nlit == avail) continue;
        // a flag byte (:1125-1186)
        const uint32_t h = knz_lzp_hash(ctx);
        const long long ref = hashes[h];
        wave_sync();
        if (lane == 0) hashes[h] = (int)dstIdx;
        bool literalFlag = ref == 0;
        if (!literalFlag) {
            srcIdx++;
            if (srcIdx >= srcEnd) { bad = true; break; }                // src[srcIdx] out of range = Go panic
            if (src[srcIdx] == 0xFF) literalFlag = true;
        }
        if (literalFlag) {
            if (dstIdx >= dstEnd) { bad = true; break; }
            if (lane == 0) dst[dstIdx] = (uint8_t)KNZ_LZP_FLAG;
            ctx = (ctx << 8) | KNZ_LZP_FLAG;
            srcIdx++; dstIdx++;
            wave_sync();
            continue;
        }
        long long mLen = KNZ_LZP_MIN_MATCH;
        if (src[srcIdx] == 0xFE) {
            while (srcIdx < srcEnd && src[srcIdx] == 0xFE) { srcIdx++; mLen += 254; }
            if (srcIdx >= srcEnd) { bad = true; break; }
        }
        mLen += src[srcIdx];
        srcIdx++;
        const long long mEnd = dstIdx + mLen;
        if (mEnd > dstEnd) { bad = true; break; }
        const long long dist = dstIdx - ref;
        if (dist <= 0) { bad = true; break; }                           // (cannot happen: the table holds earlier positions)
        wave_sync();
        __threadfence();                                                // the source bytes may have been stored by other lanes
        for (long long j = lane; j < mLen; j += 64) dst[dstIdx + j] = dst[ref + (dist >= mLen ? j : j % dist)];
        dstIdx = mEnd;
        wave_sync();
        __threadfence();
        ctx = knz_le32(dst + dstIdx - 4);
    }
    if (!bad && srcIdx != srcEnd) bad = true;
    if (lane == 0) { a.ok[b] = bad ? -KNZ_ERR_PROCESS_BLOCK : 1; a.out_len[b] = bad ? 0 : (uint32_t)dstIdx; }
}
