// Byte transforms of the hot path on gfx950: SBRT (RANK / MTFT) and ZRLT, forward and inverse, batched over the blocks
// of a stream. Replaces SBRT.Forward/Inverse (v2/transform/SBRT.go:127-175,180-226) and ZRLT.Forward/Inverse
// (v2/transform/ZRLT.go:58-137,142-225) inside ByteTransformSequence (v2/transform/Sequence.go:64-186).
//
// Forward transforms are segment-parallel:
//   RANK/MTFT: the list-update state at any position only depends on the last two accesses of every symbol, so every
//              8 KiB segment finds its symbols' last two positions (parallel), a per-block pass carries them across
//              segments, and then all segments replay the list update independently from the reconstructed state.
//   ZRLT:      element sizes (literal 1|2 bytes, zero run of k -> floor(log2(k+1)) bytes at the run's last zero) are
//              scanned per segment and per block, then scattered; the "must not expand" rule becomes a test on the total.
// Inverse transforms are serial by construction of the format (one chain per block), staged through LDS tiles so that
// global memory is only touched with coalesced accesses.
#include "bits.h"

#define KNZ_SEG 8192

struct XfArgs {
    uint32_t nblocks;
    uint32_t segs_per_block;
    const uint64_t* in_ptr;       // [nblocks] device address of the block's current bytes
    const uint32_t* in_len;       // [nblocks]
    const uint64_t* out_ptr;      // [nblocks] where this stage writes
    uint32_t out_cap;             // bytes available per block at out_ptr
    uint32_t* out_len;            // [nblocks]
    int32_t* ok;                  // [nblocks] 1 applied, 0 declined (transform skipped), <0: -(kanzi error)
    const uint8_t* active;        // [nblocks] 0 = block does not run this stage (copy block / earlier failure)
    int32_t* seg_a;               // per-segment scratch (meaning depends on the transform)
    int32_t* seg_b;
    uint32_t mode;                // SBRT mode: 1 MTF, 2 RANK, 3 TIMESTAMP
};

// After a stage: applied blocks switch to the stage output and clear their skip bit (Sequence.go:93-95)
struct CommitArgs {
    uint32_t nblocks; uint32_t stage;
    const int32_t* ok; const uint32_t* out_len; const uint64_t* out_ptr;
    uint64_t* cur_ptr; uint32_t* cur_len; uint8_t* skip; uint8_t* side; uint8_t target_if_side1, target_else;
    uint64_t base1, base2, stride; int32_t* blk_status;
};
__global__ void knz_xf_commit_kernel(CommitArgs a) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks) return;
    const int32_t ok = a.ok[b];
    if (ok == 1) {
        a.cur_ptr[b] = a.out_ptr[b];
        a.cur_len[b] = a.out_len[b];
        a.skip[b] &= (uint8_t)~(1u << (7 - a.stage));
        a.side[b] = a.side[b] == 1 ? 2 : 1;
    } else if (ok < 0) {
        a.blk_status[b] = -ok;
    }
}
// Before a stage: out_ptr = the region the block is not currently in
__global__ void knz_xf_prepare_kernel(CommitArgs a, uint64_t* out_ptr, int32_t* ok) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks) return;
    out_ptr[b] = (a.side[b] == 1 ? a.base2 : a.base1) + (uint64_t)b * a.stride;
    ok[b] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// SBRT forward
__host__ __device__ __forceinline__ int knz_sbrt_q(uint32_t mode, int i, int p) {
    // qc = ((i & mask1) + (p[c] & mask2)) >> shift   (SBRT.go:59-76,158)
    const int m1 = mode == 3 ? 0 : -1, m2 = mode == 1 ? 0 : -1, s = mode == 2 ? 1 : 0;
    return ((i & m1) + (p & m2)) >> s;
}

// The SBRT list (256 symbols ordered by rank) lives in REGISTERS: register k of lane l holds symbol, q and p of the symbol
// at rank 64*k + l. The list is always sorted by q (descending; q never decreases on an access, SBRT.go:158-168), so the
// accessed symbol lands at rank j = #{x : q[x] > qc} and one update is: a scalar look-up of the old entry (v_readlane), one
// wave-wide compare + popcount for j, and a one-lane DPP shift of the ranks j..r-1 -- a cost that does not depend on how far
// the symbol moves (the scalar loop of the reference costs two dependent memory reads per position moved). After a BWT the
// ranks are tiny, so nearly every access stays inside register 0 (ranks 0..63), and rank 0 only rewrites q/p of lane 0.
// All arguments of the methods are wave-uniform (SGPR values): the control flow is scalar branches, never divergence.
template <int MODE>
struct SbrtWave {
    uint32_t s[4];
    int q[4], p[4];

    static __device__ __forceinline__ int qf(int i, int pc) { return MODE == 1 ? i : (MODE == 2 ? ((i + pc) >> 1) : pc); }

    __device__ __forceinline__ void init_identity(int lane) {
#pragma unroll
        for (int k = 0; k < 4; k++) { s[k] = 64u * (uint32_t)k + (uint32_t)lane; q[k] = 0; p[k] = 0; }
    }
    __device__ __forceinline__ void load(const uint8_t* r2s, const int* qBySym, const int* pBySym, int lane) {
#pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t c = r2s[64 * k + lane]; s[k] = c; q[k] = qBySym[c]; p[k] = pBySym[c]; }
    }
    __device__ __forceinline__ uint32_t top() const { return wave_bcast(s[0], 0); }
    // `count` >= 1 consecutive accesses of the symbol at rank 0, the last one at time i: it stays on top (nothing above it)
    __device__ __forceinline__ void touch_top(int i, uint32_t count) {
        const int pc = count > 1 ? i - 1 : (int)wave_bcast((uint32_t)p[0], 0);
        q[0] = (int)wave_writelane0((uint32_t)q[0], (uint32_t)qf(i, pc));
        p[0] = (int)wave_writelane0((uint32_t)p[0], (uint32_t)i);
    }
    // access at time i of symbol c sitting at rank r, 0 <= r <= 63 (branch-free)
    __device__ __forceinline__ void access_low(int i, uint32_t c, uint32_t r, int lane) {
        const int qc = qf(i, (int)wave_readlane((uint32_t)p[0], r));
        const uint32_t j = (uint32_t)__popcll(wave_ballot(q[0] > qc));      // sorted list: everything with q > qc sits above r
        const bool moved = (uint32_t)lane - j - 1u < r - j;                   // j < lane <= r: one rank down
        const bool ins = (uint32_t)lane == j;
        const uint32_t sp = wave_shr1(s[0]), qp = wave_shr1((uint32_t)q[0]), pp = wave_shr1((uint32_t)p[0]);
        s[0] = ins ? c : (moved ? sp : s[0]);
        q[0] = ins ? qc : (moved ? (int)qp : q[0]);
        p[0] = ins ? i : (moved ? (int)pp : p[0]);
    }
    // rare: the symbol comes from rank 64+
    __device__ __forceinline__ void access_high(int i, uint32_t c, uint32_t r, int lane) {
        const uint32_t kr = r >> 6, l = r & 63;
        const int pc = (int)(kr == 1 ? wave_readlane((uint32_t)p[1], l) : (kr == 2 ? wave_readlane((uint32_t)p[2], l) : wave_readlane((uint32_t)p[3], l)));
        const int qc = qf(i, pc);
        uint32_t j = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) j += (uint32_t)__popcll(wave_ballot(q[k] > qc));
#pragma unroll
        for (int k = 3; k >= 0; k--) {                                            // high registers first: register k-1 is still old
            const uint32_t x = 64u * (uint32_t)k + (uint32_t)lane;
            const bool moved = x - j - 1u < r - j;
            const bool ins = x == j;
            uint32_t sp = wave_shr1(s[k]), qp = wave_shr1((uint32_t)q[k]), pp = wave_shr1((uint32_t)p[k]);
            if (k > 0) {
                const uint32_t s63 = wave_bcast(s[k > 0 ? k - 1 : 0], 63), q63 = wave_bcast((uint32_t)q[k > 0 ? k - 1 : 0], 63), p63 = wave_bcast((uint32_t)p[k > 0 ? k - 1 : 0], 63);
                if (lane == 0) { sp = s63; qp = q63; pp = p63; }
            }
            s[k] = ins ? c : (moved ? sp : s[k]);
            q[k] = ins ? qc : (moved ? (int)qp : q[k]);
            p[k] = ins ? i : (moved ? (int)pp : p[k]);
        }
    }
    __device__ __forceinline__ uint32_t find_high(uint32_t c) const {
        uint64_t m = wave_ballot(s[1] == c);
        if (m) return 64u + (uint32_t)(__ffsll((unsigned long long)m) - 1);
        m = wave_ballot(s[2] == c);
        if (m) return 128u + (uint32_t)(__ffsll((unsigned long long)m) - 1);
        m = wave_ballot(s[3] == c);
        return 192u + (uint32_t)(__ffsll((unsigned long long)m) - 1);
    }
    __device__ __forceinline__ uint32_t at_high(uint32_t r) const {
        const uint32_t k = r >> 6, l = r & 63;
        if (k == 1) return wave_readlane(s[1], l);
        if (k == 2) return wave_readlane(s[2], l);
        return wave_readlane(s[3], l);
    }
    // one symbol through whichever path it needs (used off the hot loop only)
    template <bool FWD>
    __device__ __forceinline__ uint32_t step_any(uint32_t v, int t, int lane) {
        uint32_t o;
        if (FWD) {
            const uint64_t hit = wave_ballot(s[0] == v);
            if (hit) { o = (uint32_t)(__ffsll((unsigned long long)hit) - 1); access_low(t, v, o, lane); }
            else { o = find_high(v); access_high(t, v, o, lane); }
        } else {
            if (v < 64) { o = wave_readlane(s[0], v); access_low(t, o, v, lane); }
            else { o = at_high(v); access_high(t, o, v, lane); }
        }
        return o;
    }
    // 4096-byte LDS tile, walked 256 symbols per pass: lane l holds input word l of the pass and v_readlane feeds the
    // scalar walk. Runs of words made of rank 0 only (forward: of the symbol on top) are found with one ballot per pass and
    // cost O(1) each; every other symbol takes the branch-free access_low path. The hot loop works on whole words and is
    // left (flag `stop`) when a symbol needs registers 1..3, so that the rare path adds no register shuffling to it.
    template <bool FWD>
    __device__ __forceinline__ void run_tile(const uint8_t* s_in, uint8_t* s_out, uint32_t cnt, int i0, int lane) {
        for (uint32_t base = 0; base < cnt; base += 256) {
            const uint32_t inw = *(const uint32_t*)(s_in + base + 4 * lane);
            const uint32_t m = min(256u, cnt - base);                  // symbols of this pass
            const uint32_t fullWords = m >> 2;
            uint32_t wi = 0;
            for (;;) {
                uint32_t stop = 4;                                     // index of the symbol that stopped the hot loop
                while (wi < fullWords) {
                    // words equal to 4 x (rank 0 | top symbol), starting at word wi
                    const uint32_t pat = FWD ? top() * 0x01010101u : 0u;
                    const uint64_t zm = wave_ballot(inw == pat && (uint32_t)lane < fullWords) >> wi;
                    const uint32_t run = zm == ~0ull ? 64u : (uint32_t)(__ffsll((unsigned long long)~zm) - 1);
                    if (run) {
                        touch_top(i0 + (int)(base + 4 * (wi + run) - 1), 4 * run);
                        const uint32_t fill = FWD ? 0u : top() * 0x01010101u;
                        if ((uint32_t)lane - wi < run) *(uint32_t*)(s_out + base + 4 * lane) = fill;
                        wi += run;
                        continue;
                    }
                    const uint32_t four = wave_readlane(inw, wi);
                    const int t = i0 + (int)(base + 4 * wi);
                    uint32_t outw = 0;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (stop == 4) {
                            const uint32_t v = (four >> (8 * u)) & 0xFF;
                            uint32_t o;
                            bool ok;
                            uint64_t hit = 0;
                            if (FWD) { hit = wave_ballot(s[0] == v); ok = hit != 0; } else ok = v < 64;
                            if (!ok) stop = (uint32_t)u;
                            else {
                                if (FWD) { o = (uint32_t)(__ffsll((unsigned long long)hit) - 1); access_low(t + u, v, o, lane); }
                                else { o = wave_readlane(s[0], v); access_low(t + u, o, v, lane); }
                                outw |= o << (8 * u);
                            }
                        }
                    }
                    *(uint32_t*)(s_out + base + 4 * wi) = outw;        // same word from every lane
                    if (stop != 4) break;
                    wi++;
                }
                if (stop == 4) break;
                const uint32_t four = wave_readlane(inw, wi);
                for (uint32_t u = stop; u < 4; u++)
                    s_out[base + 4 * wi + u] = (uint8_t)step_any<FWD>((four >> (8 * u)) & 0xFF, i0 + (int)(base + 4 * wi + u), lane);
                wi++;
            }
            if (m & 3) {
                const uint32_t four = wave_readlane(inw, fullWords);
                for (uint32_t u = 0; u < (m & 3); u++)
                    s_out[base + 4 * fullWords + u] = (uint8_t)step_any<FWD>((four >> (8 * u)) & 0xFF, i0 + (int)(base + 4 * fullWords + u), lane);
            }
        }
    }
};

// 1) last two positions of every symbol inside the segment (block-local positions, -1 if none)
__global__ __launch_bounds__(64) void knz_sbrt_seg_last2_kernel(XfArgs a) {
    __shared__ int s_last[256], s_prev[256];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    int32_t* oa = a.seg_a + (size_t)blockIdx.x * 256;
    int32_t* ob = a.seg_b + (size_t)blockIdx.x * 256;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG);
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    for (int i = lane; i < 256; i += 64) { s_last[i] = -1; s_prev[i] = -1; }
    wave_sync();
    // A lane takes 16 consecutive positions per trip (one 16-byte read) and leaves out the positions that cannot be an answer: one followed by
    // the same symbol is not its last occurrence, one followed by it twice is not the last but one either (behind a BWT most positions are
    // inside a run: a byte per lane and trip with every position an LDS atomic on the same counter was 0.96 ms for 212 MB).
    for (int pass = 0; pass < 2; pass++) {
        for (uint32_t p0 = lo + 16u * lane; p0 < hi; p0 += 16u * 64u) {
            const uint32_t cnt = min(16u, hi - p0);
            uint32_t w[4] = {0u, 0u, 0u, 0u};
            if (cnt == 16u) { const uint64_t a0 = knz_vle64(src + p0), a1 = knz_vle64(src + p0 + 8); w[0] = (uint32_t)a0; w[1] = (uint32_t)(a0 >> 32); w[2] = (uint32_t)a1; w[3] = (uint32_t)(a1 >> 32); }
            else for (uint32_t j = 0; j < cnt; j++) w[j >> 2] |= (uint32_t)src[p0 + j] << (8u * (j & 3u));
#pragma unroll
            for (uint32_t j = 0; j < 16u; j++) {
                if (j >= cnt) break;
                const uint32_t c = (w[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
                const bool same1 = j + 1u < cnt && ((w[(j + 1u) >> 2] >> (8u * ((j + 1u) & 3u))) & 0xFFu) == c;
                const bool same2 = j + 2u < cnt && ((w[(j + 2u) >> 2] >> (8u * ((j + 2u) & 3u))) & 0xFFu) == c;
                const int i = (int)(p0 + j);
                if (pass == 0) { if (!same1) atomicMax(&s_last[c], i); }
                else if (!(same1 && same2)) { if (i != s_last[c]) atomicMax(&s_prev[c], i); }
            }
        }
        wave_sync();
    }
    for (int i = lane; i < 256; i += 64) { oa[i] = s_last[i]; ob[i] = s_prev[i]; }
}

// 2) carry across the segments of a block: on exit seg_a/seg_b hold (last, prev) access of each symbol BEFORE the segment
__global__ __launch_bounds__(256) void knz_sbrt_carry_kernel(XfArgs a) {
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const int d = threadIdx.x;
    const uint32_t n = a.in_len[b];
    const uint32_t nseg = (n + KNZ_SEG - 1) / KNZ_SEG;
    int last = -1, prev = -1;
    int32_t* pa = a.seg_a + (size_t)b * a.segs_per_block * 256 + d;
    int32_t* pb = a.seg_b + (size_t)b * a.segs_per_block * 256 + d;
    for (uint32_t s0 = 0; s0 < nseg; s0 += 8) {                             // (eight segments' reads in flight: the in-place update read and wrote one segment per memory latency)
        int li[8], pi[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) { const bool in = s0 + k < nseg; li[k] = in ? pa[(size_t)(s0 + k) * 256] : -1; pi[k] = in ? pb[(size_t)(s0 + k) * 256] : -1; }
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) {
            if (s0 + k >= nseg) break;
            pa[(size_t)(s0 + k) * 256] = last; pb[(size_t)(s0 + k) * 256] = prev;
            if (li[k] >= 0) { prev = pi[k] >= 0 ? pi[k] : last; last = li[k]; }
        }
    }
}

// 3) replay of the list update inside each segment from the reconstructed state (SBRT.go:155-172)
#define KNZ_SBRT_TILE 2048
template <int MODE>
__global__ __launch_bounds__(64) void knz_sbrt_apply_kernel(XfArgs a) {
    // 4.5 KB of LDS per wave: the segment is walked in tiles of 2 KiB and the tables that rebuild the list alias the tile buffers (round 4: the
    // whole 8 KiB segment + its output + the tables were 20 KB = 2 waves per SIMD; the replay is instruction issue of one wave per segment, so
    // the waves a SIMD can interleave set its rate)
    __shared__ __attribute__((aligned(16))) uint8_t s_buf[2 * KNZ_SBRT_TILE];
    __shared__ uint8_t s_s2r[256], s_r2s[256];
    uint8_t* s_in = s_buf; uint8_t* s_out = s_buf + KNZ_SBRT_TILE;
    int* s_p = (int*)s_buf; int* s_q = s_p + 256; int* s_t = s_q + 256;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    if (s == 0 && lane == 0) { a.out_len[b] = n; a.ok[b] = (a.out_cap >= n + 33) ? 1 : 0; }   // MaxEncodedLen check (:137-139)
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG), cnt = hi - lo;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const int32_t* ca = a.seg_a + (size_t)blockIdx.x * 256;
    const int32_t* cb = a.seg_b + (size_t)blockIdx.x * 256;
    for (int d = lane; d < 256; d += 64) {
        const int last = ca[d], prev = cb[d];
        s_t[d] = last;                                            // -1 = never accessed
        s_p[d] = last >= 0 ? last : 0;                            // p[] starts at 0 (:150)
        s_q[d] = last >= 0 ? knz_sbrt_q(a.mode, last, prev >= 0 ? prev : 0) : 0;
    }
    wave_sync();
    // list order = (q desc, more recently accessed first, never accessed in symbol order)
    for (int d = lane; d < 256; d += 64) {
        const int qd = s_q[d], td = s_t[d];
        int r = 0;
        for (int e = 0; e < 256; e++) {
            const int qe = s_q[e], te = s_t[e];
            const bool above = qe > qd || (qe == qd && (te > td || (te == td && e < d)));
            r += above ? 1 : 0;
        }
        s_s2r[d] = (uint8_t)r;
        s_r2s[r] = (uint8_t)d;
    }
    wave_sync();
    SbrtWave<MODE> w;
    w.load(s_r2s, s_q, s_p, lane);
    for (uint32_t base = 0; base < cnt; base += KNZ_SBRT_TILE) {
        const uint32_t tc = min((uint32_t)KNZ_SBRT_TILE, cnt - base);
        wave_sync();                                              // (the tables, then the tile before, have been read)
        for (uint32_t i = lane; i < tc; i += 64) s_in[i] = src[lo + base + i];
        wave_sync();
        w.template run_tile<true>(s_in, s_out, tc, (int)(lo + base), lane);
        wave_sync();
        for (uint32_t i = lane; i < tc; i += 64) dst[lo + base + i] = s_out[i];
    }
}

// SBRT inverse: one chain per block (SBRT.go:204-223), tiles staged through LDS
template <int MODE>
__global__ __launch_bounds__(64) void knz_sbrt_inverse_kernel(XfArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[4096];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[4096];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    if (lane == 0) { a.out_len[b] = n; a.ok[b] = n <= a.out_cap ? 1 : -KNZ_ERR_PROCESS_BLOCK; }
    if (n > a.out_cap) return;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    SbrtWave<MODE> w;
    w.init_identity(lane);
    for (uint32_t base = 0; base < n; base += 4096) {
        const uint32_t cnt = min(4096u, n - base);
        wave_sync();
        for (uint32_t i = lane; i < cnt; i += 64) s_in[i] = src[base + i];
        wave_sync();
        w.template run_tile<false>(s_in, s_out, cnt, (int)base, lane);
        wave_sync();
        for (uint32_t i = lane; i < cnt; i += 64) dst[base + i] = s_out[i];
    }
}

// MTFT inverse without the block-long chain. A move-to-front step permutes list POSITIONS in a way that does not depend on
// what the list holds (rank r goes to the front, ranks 0..r-1 move down): the effect of a whole segment of ranks on the list
// is a permutation that can be worked out from the ranks alone. So: (1) every segment runs its ranks over the identity list
// and keeps the permutation it ends with, (2) one workgroup per block composes the permutations in order, which gives the
// list at the start of every segment, (3) every segment decodes from its own start list. Two 8192-step chains per segment,
// all segments at once, instead of one chain over the block. (RANK has no such form: where a symbol lands depends on the
// access times stored with the list entries, which only the preceding ranks can tell.)
__global__ __launch_bounds__(64) void knz_mtft_inv_perm_kernel(XfArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[KNZ_SEG];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[KNZ_SEG];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    if (lo >= n || n > a.out_cap) return;
    const uint32_t cnt = min(n, lo + KNZ_SEG) - lo;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    for (uint32_t i = lane; i < cnt; i += 64) s_in[i] = src[lo + i];
    wave_sync();
    SbrtWave<1> w;
    w.init_identity(lane);
    w.template run_tile<false>(s_in, s_out, cnt, 1, lane);                         // (times start at 1: the fresh list holds q = 0)
    uint8_t* perm = (uint8_t*)(a.seg_a + (size_t)blockIdx.x * 256);               // 256 bytes of the segment's 1 KiB slot
#pragma unroll
    for (int k = 0; k < 4; k++) perm[64 * k + lane] = (uint8_t)w.s[k];            // position j of the list now holds what was at perm[j]
}

__global__ __launch_bounds__(256) void knz_mtft_inv_compose_kernel(XfArgs a) {
    __shared__ uint8_t s_cur[256], s_nxt[256];
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const int j = threadIdx.x;
    const uint32_t n = a.in_len[b];
    if (n > a.out_cap) return;
    const uint32_t nseg = (n + KNZ_SEG - 1) / KNZ_SEG;
    s_cur[j] = (uint8_t)j;                                                          // SBRT starts from the identity list (:186-190)
    __syncthreads();
    for (uint32_t s = 0; s < nseg; s++) {
        const uint8_t* perm = (const uint8_t*)(a.seg_a + ((size_t)b * a.segs_per_block + s) * 256);
        uint8_t* start = (uint8_t*)(a.seg_b + ((size_t)b * a.segs_per_block + s) * 256);
        start[j] = s_cur[j];
        s_nxt[j] = s_cur[perm[j]];
        __syncthreads();
        s_cur[j] = s_nxt[j];
        __syncthreads();
    }
}

__global__ __launch_bounds__(64) void knz_mtft_inv_apply_kernel(XfArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[KNZ_SEG];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[KNZ_SEG];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    if (s == 0 && lane == 0) { a.out_len[b] = n; a.ok[b] = n <= a.out_cap ? 1 : -KNZ_ERR_PROCESS_BLOCK; }
    const uint32_t lo = s * KNZ_SEG;
    if (lo >= n || n > a.out_cap) return;
    const uint32_t cnt = min(n, lo + KNZ_SEG) - lo;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const uint8_t* start = (const uint8_t*)(a.seg_b + (size_t)blockIdx.x * 256);
    for (uint32_t i = lane; i < cnt; i += 64) s_in[i] = src[lo + i];
    SbrtWave<1> w;
#pragma unroll
    for (int k = 0; k < 4; k++) { w.s[k] = start[64 * k + lane]; w.q[k] = 0; w.p[k] = 0; }
    wave_sync();
    w.template run_tile<false>(s_in, s_out, cnt, 1, lane);
    wave_sync();
    for (uint32_t i = lane; i < cnt; i += 64) dst[lo + i] = s_out[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// ZRLT forward
// 1) per segment: block-local index of the last non-zero byte (-1 if none)
__global__ __launch_bounds__(256) void knz_zrlt_seg_lastnz_kernel(XfArgs a) {
    __shared__ int s_max;
    const int tid = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG);
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    if (tid == 0) s_max = -1;
    __syncthreads();
    int m = -1;
    for (uint32_t i = lo + tid; i < hi; i += 256) if (src[i] != 0) m = (int)i;
    if (m >= 0) atomicMax(&s_max, m);
    __syncthreads();
    if (tid == 0) a.seg_a[blockIdx.x] = s_max;
}

// element size at position i given the index of the last non-zero byte before it (ZRLT.go:76-124)
__device__ __forceinline__ uint32_t knz_zrlt_elem_size(const uint8_t* src, uint32_t n, uint32_t i, int lastNZ) {
    const uint8_t v = src[i];
    if (v != 0) return v >= 0xFE ? 2u : 1u;
    if (i + 1 < n && src[i + 1] == 0) return 0u;           // not the last zero of its run
    const uint32_t k = (uint32_t)((int)i - lastNZ);        // run length
    return 31u - (uint32_t)__builtin_clz(k + 1);           // floor(log2(k+1)) digit bytes
}

// 2) per block: exclusive max-scan of the segments' last non-zero index -> seg_a[s] = last non-zero BEFORE segment s
__global__ __launch_bounds__(64) void knz_zrlt_carry_kernel(XfArgs a) {
    const uint32_t b = blockIdx.x;
    if (!a.active[b] || threadIdx.x != 0) return;
    const uint32_t nseg = (a.in_len[b] + KNZ_SEG - 1) / KNZ_SEG;
    int carry = -1;
    for (uint32_t s = 0; s < nseg; s++) {
        int32_t* p = a.seg_a + (size_t)b * a.segs_per_block + s;
        const int v = *p;
        *p = carry;
        if (v > carry) carry = v;
    }
}

// 3) per segment: output bytes produced by the elements that END in this segment -> seg_b
// 5) (scatter = true) write them at seg_b (now the segment's output offset)
template <bool SCATTER>
__global__ __launch_bounds__(256) void knz_zrlt_seg_kernel(XfArgs a) {
    // Round 4: the segment (8 KiB = 256 threads x 32 consecutive bytes) is staged in LDS with 16-byte loads and its output leaves through LDS as
    // whole words: the thread-per-32-bytes loops used to read and write global memory a byte at a time (64 lanes = 64 different 32-byte
    // stretches per load instruction), which kept both passes at 3 % of the HBM rate.
    __shared__ __attribute__((aligned(16))) uint8_t s_in[KNZ_SEG + 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_o[2 * KNZ_SEG + 64];                    // (8192 literals of two bytes, or a few less behind the digits of a run that ends here)
    __shared__ uint32_t s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    if (SCATTER && a.ok[b] != 1) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG), cnt = hi - lo;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const uint32_t carry0 = SCATTER ? (uint32_t)a.seg_b[blockIdx.x] : 0u;
    const int lastnz0 = a.seg_a[blockIdx.x];
    if ((((uintptr_t)(src + lo)) & 15) == 0) {
        for (uint32_t i = (uint32_t)tid * 16; i < cnt; i += 256 * 16) {
            if (i + 16 <= cnt) *(uint4*)(s_in + i) = *(const uint4*)(src + lo + i);
            else for (uint32_t k = i; k < cnt; k++) s_in[k] = src[lo + k];
        }
    } else {
        for (uint32_t i = tid; i < cnt; i += 256) s_in[i] = src[lo + i];
    }
    if (tid == 0) s_in[cnt] = hi < n ? src[hi] : (uint8_t)1;                // the byte behind the segment (a zero run may go on there); behind the block: "not zero"
    __syncthreads();
    const uint32_t x0 = (uint32_t)tid * 32, p0 = lo + x0;
    const uint32_t mine = x0 < cnt ? min(32u, cnt - x0) : 0u;
    // last non-zero before p0: scan of the per-thread maxima of the previous threads
    int myMax = -1;
    for (uint32_t j = 0; j < mine; j++) if (s_in[x0 + j] != 0) myMax = (int)(p0 + j);
    int m = myMax;
    for (int d = 1; d < 64; d <<= 1) { int t = (int)wave_shfl((uint32_t)m, lane - d); if (lane >= d && t > m) m = t; }
    if (lane == 63) s_wave[wave] = (uint32_t)m;
    __syncthreads();
    int before = lastnz0;
    for (int w = 0; w < wave; w++) { int t = (int)s_wave[w]; if (t > before) before = t; }
    int exclusive = (int)wave_shfl((uint32_t)m, lane - 1);
    if (lane == 0) exclusive = -1;
    if (exclusive > before) before = exclusive;
    // sizes of my elements (element = literal, or a zero run accounted at its last zero: ZRLT.go:76-124)
    uint32_t sz = 0;
    int lnz = before;
    for (uint32_t j = 0; j < mine; j++) {
        const uint8_t v = s_in[x0 + j];
        if (v != 0) { sz += v >= 0xFE ? 2u : 1u; lnz = (int)(p0 + j); }
        else if (s_in[x0 + j + 1] != 0) sz += 31u - (uint32_t)__builtin_clz((uint32_t)((int)(p0 + j) - lnz) + 1);
    }
    const uint32_t incl = wave_scan_incl(sz);
    __syncthreads();
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t off = incl - sz;                                               // inside the segment's output
    for (int w = 0; w < wave; w++) off += s_wave[w];
    const uint32_t total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (!SCATTER) { if (tid == 0) a.seg_b[blockIdx.x] = (int32_t)total; return; }
    lnz = before;
    for (uint32_t j = 0; j < mine; j++) {
        const uint32_t i = p0 + j;
        const uint8_t v = s_in[x0 + j];
        if (v != 0) {
            if (v >= 0xFE) { s_o[off++] = 0xFF; s_o[off++] = (uint8_t)(v - 0xFE); }
            else s_o[off++] = (uint8_t)(v + 1);
            lnz = (int)i;
        } else if (s_in[x0 + j + 1] != 0) {
            const uint32_t run = (uint32_t)((int)i - lnz) + 1;              // runLength = k + 1 (:88)
            uint32_t lg = 31u - (uint32_t)__builtin_clz(run);
            while (lg > 0) { lg--; s_o[off++] = (uint8_t)((run >> lg) & 1); }
        }
    }
    __syncthreads();
    // the segment's output: bytes up to the first 4-byte boundary of the destination, whole words, the rest
    uint8_t* out = dst + carry0;
    const uint32_t head = min(total, (uint32_t)((4 - ((uintptr_t)out & 3)) & 3));
    if ((uint32_t)tid < head) out[tid] = s_o[tid];
    const uint32_t words = (total - head) >> 2;
    for (uint32_t wq = tid; wq < words; wq += 256) {
        const uint8_t* q = s_o + head + 4 * wq;
        *(uint32_t*)(out + head + 4 * wq) = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
    }
    const uint32_t done = head + 4 * words;
    if ((uint32_t)tid < total - done) out[done + tid] = s_o[done + tid];
}

// 4) per block: exclusive scan of the segment sizes, total, and the "would not fit in len(src)" test (ZRLT.go:93,109,118,132):
//    every write needs dstIdx < len(src) (a zero run even dstIdx + digits < len(src)), i.e. total < n, or total == n
//    when the last element is a literal.
__global__ __launch_bounds__(64) void knz_zrlt_offsets_kernel(XfArgs a) {
    const uint32_t b = blockIdx.x;
    if (!a.active[b] || threadIdx.x != 0) return;
    const uint32_t n = a.in_len[b];
    const uint32_t nseg = (n + KNZ_SEG - 1) / KNZ_SEG;
    uint64_t off = 0;
    for (uint32_t s = 0; s < nseg; s++) {
        int32_t* p = a.seg_b + (size_t)b * a.segs_per_block + s;
        const uint32_t v = (uint32_t)*p;
        *p = (int32_t)off;
        off += v;
    }
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    const bool fits = off < n || (off == n && src[n - 1] != 0);
    a.out_len[b] = (uint32_t)off;
    a.ok[b] = (fits && a.out_cap >= n) ? 1 : 0;
}

// ZRLT inverse (ZRLT.go:142-225), segment-parallel like the forward direction. Every input byte is classified
// (payload of a 0xFF escape / digit of a zero run / escape / literal); a run of digit bytes d1..dL decodes to
// ((1<<L)|d1..dL) - 1 zeros, accounted at its last digit; literals and payloads produce one byte. Output offsets come
// from a per-segment + per-block scan, the destination is zero-filled and only non-zero bytes are scattered.
// class: 0 literal, 1 digit, 2 escape (0xFF), 3 payload
__device__ __forceinline__ int knz_zi_class(const uint8_t* src, uint32_t i) {
    uint32_t c = 0;
    while (c < i && c < 4096 && src[i - 1 - c] == 0xFF) c++;     // valid streams: 0 or 1 (an escape's payload is 0 or 1)
    if (c & 1) return 3;
    const uint8_t v = src[i];
    return v <= 1 ? 1 : (v == 0xFF ? 2 : 0);
}
// bytes produced by the element ending at i ; lastND = index of the last non-digit byte before i (-1 if none)
__device__ __forceinline__ uint32_t knz_zi_size(const uint8_t* src, uint32_t n, uint32_t i, int cls, int lastND, int* err) {
    if (cls == 0 || cls == 3) return 1;
    if (cls == 2) return 0;
    if (i + 1 < n && knz_zi_class(src, i + 1) == 1) return 0;       // not the last digit of its run
    const uint32_t L = (uint32_t)((int)i - lastND);
    if (L > 31) { *err = 1; return 0; }
    uint32_t v = 1;
    for (uint32_t j = 0; j < L; j++) v = (v << 1) | src[i - L + 1 + j];
    return v - 1;
}

// 1) per segment: block-local index of the last NON-digit byte (-1 if none)
__global__ __launch_bounds__(256) void knz_zrlti_seg_lastnd_kernel(XfArgs a) {
    __shared__ int s_max;
    const int tid = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG);
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    if (tid == 0) s_max = -1;
    __syncthreads();
    int m = -1;
    for (uint32_t i = lo + tid; i < hi; i += 256) if (knz_zi_class(src, i) != 1) m = (int)i;
    if (m >= 0) atomicMax(&s_max, m);
    __syncthreads();
    if (tid == 0) a.seg_a[blockIdx.x] = s_max;
}

template <bool SCATTER>
__global__ __launch_bounds__(256) void knz_zrlti_seg_kernel(XfArgs a) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    __shared__ int s_lastnd;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    if (SCATTER && a.ok[b] != 1) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG);
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    if (tid == 0) { s_carry = SCATTER ? (uint32_t)a.seg_b[blockIdx.x] : 0u; s_lastnd = a.seg_a[blockIdx.x]; }
    __syncthreads();
    int err = 0;
    for (uint32_t base = lo; base < hi; base += 256 * 32) {
        const uint32_t p0 = base + (uint32_t)tid * 32;
        int myMax = -1;
        for (uint32_t j = 0; j < 32 && p0 + j < hi; j++) if (knz_zi_class(src, p0 + j) != 1) myMax = (int)(p0 + j);
        int m = myMax;
        for (int d = 1; d < 64; d <<= 1) { int t = (int)wave_shfl((uint32_t)m, lane - d); if (lane >= d && t > m) m = t; }
        __syncthreads();
        if (lane == 63) s_wave[wave] = (uint32_t)m;
        __syncthreads();
        int before = s_lastnd;
        for (int w = 0; w < wave; w++) { int t = (int)s_wave[w]; if (t > before) before = t; }
        int exclusive = (int)wave_shfl((uint32_t)m, lane - 1);
        if (lane == 0) exclusive = -1;
        if (exclusive > before) before = exclusive;
        int passMax = (int)s_wave[0];
        for (int w = 1; w < 4; w++) if ((int)s_wave[w] > passMax) passMax = (int)s_wave[w];
        uint32_t sz = 0;
        int lnd = before;
        for (uint32_t j = 0; j < 32 && p0 + j < hi; j++) {
            const int cls = knz_zi_class(src, p0 + j);
            sz += knz_zi_size(src, n, p0 + j, cls, lnd, &err);
            if (cls != 1) lnd = (int)(p0 + j);
        }
        const uint32_t incl = wave_scan_incl(sz);
        __syncthreads();
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t off = s_carry + incl - sz;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        const uint32_t passTotal = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (SCATTER) {
            lnd = before;
            for (uint32_t j = 0; j < 32 && p0 + j < hi; j++) {
                const uint32_t i = p0 + j;
                const int cls = knz_zi_class(src, i);
                const uint32_t e = knz_zi_size(src, n, i, cls, lnd, &err);
                if (cls == 0) dst[off] = (uint8_t)(src[i] - 1);
                else if (cls == 3) dst[off] = (uint8_t)(0xFE + src[i]);
                off += e;
                if (cls != 1) lnd = (int)i;
            }
        }
        __syncthreads();
        if (tid == 0) { s_carry += passTotal; if (passMax > s_lastnd) s_lastnd = passMax; }
        __syncthreads();
    }
    if (!SCATTER && tid == 0) a.seg_b[blockIdx.x] = (int32_t)s_carry;
    if (!SCATTER && err) a.ok[b] = -KNZ_ERR_PROCESS_BLOCK;
}

// per block: exclusive scan of the segment sizes; the decoded size must fit the destination (ZRLT.go:178-180,211-216)
__global__ __launch_bounds__(64) void knz_zrlti_offsets_kernel(XfArgs a) {
    const uint32_t b = blockIdx.x;
    if (!a.active[b] || threadIdx.x != 0) return;
    const uint32_t n = a.in_len[b];
    const uint32_t nseg = (n + KNZ_SEG - 1) / KNZ_SEG;
    uint64_t off = 0;
    for (uint32_t s = 0; s < nseg; s++) {
        int32_t* p = a.seg_b + (size_t)b * a.segs_per_block + s;
        const uint32_t v = (uint32_t)*p;
        *p = (int32_t)off;
        off += v;
    }
    a.out_len[b] = (uint32_t)min(off, (uint64_t)0xFFFFFFFFu);
    if (a.ok[b] == 0) a.ok[b] = off <= a.out_cap ? 1 : -KNZ_ERR_PROCESS_BLOCK;
}

// zero fill of the decoded range (only non-zero bytes are scattered afterwards)
__global__ __launch_bounds__(256) void knz_zrlti_zero_kernel(XfArgs a) {
    const uint32_t b = blockIdx.y;
    if (!a.active[b] || a.ok[b] != 1) return;
    const uint32_t n = a.out_len[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16; i < n; i += (uint64_t)gridDim.x * 256 * 16) {
        if (i + 16 <= n && (((uintptr_t)(dst + i)) & 15) == 0) { uint4 z; z.x = z.y = z.z = z.w = 0; *(uint4*)(dst + i) = z; }
        else for (uint64_t j = i; j < n && j < i + 16; j++) dst[j] = 0;
    }
}

// NullTransform inside a sequence: applies, nothing moves
__global__ void knz_xf_none_kernel(uint32_t nblocks, const uint8_t* active, uint8_t* skip, uint32_t stage) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblocks && active[b]) skip[b] &= (uint8_t)~(1u << (7 - stage));
}
// inverse sequence: a stage runs for live blocks whose skip bit is clear; `live` (take) is the decode-side block liveness
// piped: blocks whose stage the fused ZRLT / RANK inverse already ran (rank_pipe.hip), or null
__global__ void knz_xf_inv_select_kernel(uint32_t nblocks, const uint8_t* skip, const uint8_t* live, uint8_t* active, uint32_t stage, const int32_t* blk_status, const uint8_t* piped) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblocks) active[b] = (live[b] && blk_status[b] == 0 && !(skip[b] & (1u << (7 - stage))) && !(piped != nullptr && piped[b])) ? 1 : 0;
}
// gather of per-block byte ranges (final copy of decoded blocks / staging)
__global__ __launch_bounds__(256) void knz_copy_blocks_kernel(uint32_t nblocks, const uint64_t* src_ptr, const uint32_t* len, const uint64_t* dst_ptr, const uint8_t* live) {
    const uint32_t b = blockIdx.y;
    if (b >= nblocks || (live && !live[b])) return;
    const uint32_t n = len[b];
    const uint8_t* s = (const uint8_t*)src_ptr[b];
    uint8_t* d = (uint8_t*)dst_ptr[b];
    if (s == d) return;
    for (uint32_t i = (blockIdx.x * 256 + threadIdx.x) * 16; i < n; i += gridDim.x * 256 * 16) {
        if (i + 16 <= n && ((((uintptr_t)(s + i)) | ((uintptr_t)(d + i))) & 15) == 0) *(uint4*)(d + i) = *(const uint4*)(s + i);
        else for (uint32_t j = i; j < n && j < i + 16; j++) d[j] = s[j];
    }
}
