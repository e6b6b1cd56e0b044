// Forward suffix sort of the BWT stage, round-4 form (replaces DivSufSort.ComputeBWT's sorting, v2/transform/DivSufSort.go:179-525,
// for every block of a batch at once). Prefix doubling as before, but nothing that is already in order is sorted again:
//
//   round 0   per-block LSD radix sort of the first KNZ_SS_K0 symbols (blocks are contiguous segments of the suffix array, so the
//             block id is NOT part of the key: one pass fewer than sorting the concatenation); the first pass reads the text itself
//             (no key array is built first).
//   classify  equal keys = a group. Groups live as head bits; a group of one suffix is final. Unresolved suffixes go to one of two
//             lists by the size of their group: "normal" (2..KNZ_SG_T members) or "large".
//   round h   the groups of the normal list are refined by a SEGMENTED sort: knz_sg_keys_kernel gathers rank[i + h] for every entry (a streaming
//             kernel: the random gathers are what costs, they want every lane of the chip), knz_sg_sort_kernel: a workgroup takes the groups
//             that start inside its stretch of the list (< 2 * KNZ_SG_T items), sorts (group, rank) in LDS with 9-bit digits and writes the
//             suffixes back in place: one read and one write of the list per round instead of seven device-wide radix passes over
//             (group start : 28 | rank : 28) keys. Only the large groups (the most frequent words, long runs) take the device-wide radix sort,
//             on (dense group index | rank) keys; once that list is short its keys are run-aware (knz_sl_keys_kernel): a run of one symbol is
//             resolved in one round whatever its length.
//   update    knz_sg_update_kernel scatters the ranks that changed and compacts the list (every gather of a round reads the ranks of the
//             round before, so the scatter is a kernel of its own).
// rank[i] = 1 + first slot of i's group, counted from the block's first slot (block-local: 8 bits fewer to sort than global ranks).
// A suffix that ends within the next h symbols sorts in front of every member of its group that goes on (second key: its length - 1,
// which is < h; the others carry rank + h), as in rounds 1-3.
#include "bits.h"

#ifndef KNZ_SS_K0
#define KNZ_SS_K0 8                       // symbols of the first key (measured on S-silesia, docs/HISTORY.md "Suffix sort": 4, 5, 6, 7, 8)
#endif
#define KNZ_SS_MASK 0x3FFFFFFFu           // slot numbers stay below 2^30 (KNZ_BWT_GROUP_BYTES)
#define KNZ_SS_HEAD 0x80000000u           // list entry: the slot starts a group
#define KNZ_SS_SINGLE 0x80000000u         // sort result: the item's new group has one member (final)
#define KNZ_SS_SAME 0x40000000u           // sort result: the item's rank did not change
#ifndef KNZ_SG_T
#define KNZ_SG_T 2048                     // largest group of the normal list = list entries per workgroup stretch
#endif
#ifndef KNZ_SG_THREADS
#define KNZ_SG_THREADS 512
#endif

struct SsGeom {
    uint32_t nblocks;
    const uint32_t* gstart;               // [nblocks + 1] first slot / suffix index of each block (blocks that do not take part: empty)
    const uint32_t* tile_base;            // [nblocks + 1] first radix tile of each block
    const uint64_t* in_ptr;               // [nblocks]
    const uint32_t* in_len;               // [nblocks] (0: block does not take part)
};

__device__ __forceinline__ uint32_t knz_ss_block_of(const uint32_t* starts, uint32_t nblocks, uint32_t x) {
    uint32_t lo = 0, hi = nblocks;        // largest b with starts[b] <= x (the last of several equal ones: the non-empty block)
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (starts[mid] <= x) lo = mid; else hi = mid; }
    return lo;
}

// first K0 symbols of the suffix at `loc`, big-endian, zeros behind the end of the block
__device__ __forceinline__ uint64_t knz_ss_text_key(const uint8_t* src, uint32_t loc, uint32_t n) {
    uint64_t k = 0;
    if (loc + 8 <= n) {
        uint64_t v;
        __builtin_memcpy(&v, src + loc, 8);
        k = __builtin_bswap64(v) >> (64 - 8 * KNZ_SS_K0);
    } else {
#pragma unroll
        for (int j = 0; j < KNZ_SS_K0; j++) k = (k << 8) | (loc + j < n ? (uint64_t)src[loc + j] : 0);
    }
    return k;
}

// lanes of the wave whose digit (nbits wide) equals this lane's
__device__ __forceinline__ uint64_t knz_ss_match(uint32_t d, bool valid, int nbits) {
    uint64_t m = wave_ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 9; bit++) {
        if (bit < nbits) {
            const bool one = (d >> bit) & 1u;
            const uint64_t b = wave_ballot(one);
            m &= one ? b : ~b;
        }
    }
    return m;
}

template <int NW>
__device__ __forceinline__ uint32_t knz_ss_wg_scan_excl(uint32_t v, uint32_t* s_w, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_scan_incl(v);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t acc = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { const uint32_t x = s_w[k]; if ((uint32_t)k < w) acc += x; tot += x; }
    __syncthreads();
    total = tot;
    return acc + incl - v;
}
template <int NW>
__device__ __forceinline__ uint32_t knz_ss_wg_scan_incl_max(uint32_t v, uint32_t* s_w) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = wave_shfl(x, (int)((lane - (uint32_t)d) & 63u)); if ((int)lane >= d) x = x > o ? x : o; }
    if (lane == 63) s_w[w] = x;
    __syncthreads();
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { const uint32_t y = s_w[k]; if ((uint32_t)k < w) acc = acc > y ? acc : y; }
    __syncthreads();
    return x > acc ? x : acc;
}

// ---- round 0: per-block LSD radix sort (8-bit digits, tiles of 4096 = KNZ_RS_TILE, the ranking of prims.hip) ------------------------------
struct SsTile { uint32_t b, lt, tpb, n, t0, gs0; };
__device__ __forceinline__ SsTile knz_ss_tile(const SsGeom& g, uint32_t tile) {
    SsTile t;
    t.b = knz_ss_block_of(g.tile_base, g.nblocks, tile);
    t.lt = tile - g.tile_base[t.b];
    t.tpb = g.tile_base[t.b + 1] - g.tile_base[t.b];
    t.n = g.in_len[t.b];
    t.t0 = t.lt * KNZ_RS_TILE;
    t.gs0 = g.gstart[t.b];
    return t;
}

// per-tile digit counts, laid out block-major, digit-major, tile-minor: ONE flat exclusive sum then gives every (block, digit, tile) its first
// output slot with the blocks kept apart
template <bool FIRST>
__global__ __launch_bounds__(KNZ_RS_THREADS) void knz_ss_hist_kernel(SsGeom g, const uint64_t* keys, unsigned shift, uint32_t* hist) {
    // counts only (no ranks): plain LDS atomics into four copies per wave (lane & 3), which costs a few LDS cycles per row of 64 keys even when
    // many lanes share a digit; the one-ballot-per-bit matching of the scatter costs ~70 instructions per row and made this kernel compute-bound
    __shared__ uint32_t s_cnt[KNZ_RS_THREADS / 64][4][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const SsTile t = knz_ss_tile(g, blockIdx.x);
    const uint8_t* src = (const uint8_t*)g.in_ptr[t.b];
    for (uint32_t i = tid; i < (KNZ_RS_THREADS / 64) * 4 * 256; i += KNZ_RS_THREADS) (&s_cnt[0][0][0])[i] = 0;
    __syncthreads();
    uint32_t* mine = s_cnt[w][lane & 3];
#pragma unroll
    for (int r = 0; r < KNZ_RS_ITEMS; r++) {
        const uint32_t loc = t.t0 + w * (64 * KNZ_RS_ITEMS) + (uint32_t)r * 64 + lane;
        if (loc < t.n) {
            uint32_t d;
            if (FIRST) d = loc + (KNZ_SS_K0 - 1) < t.n ? src[loc + (KNZ_SS_K0 - 1)] : 0u;
            else d = (uint32_t)(keys[t.gs0 + loc] >> shift) & 0xFFu;
            atomicAdd(&mine[d], 1u);
        }
    }
    __syncthreads();
    uint32_t sum = 0;
    for (int k = 0; k < KNZ_RS_THREADS / 64; k++) sum += s_cnt[k][0][tid] + s_cnt[k][1][tid] + s_cnt[k][2][tid] + s_cnt[k][3][tid];
    hist[(size_t)256 * g.tile_base[t.b] + (size_t)tid * t.tpb + t.lt] = sum;
}

template <bool FIRST>
__global__ __launch_bounds__(KNZ_RS_THREADS) void knz_ss_scatter_kernel(SsGeom g, const uint64_t* kin, const uint32_t* vin, uint64_t* kout, uint32_t* vout,
                                                                       unsigned shift, const uint32_t* offs) {
    __shared__ uint32_t s_cnt[KNZ_RS_THREADS / 64][256];
    __shared__ uint32_t s_gbase[256];
    __shared__ uint32_t s_w[KNZ_RS_THREADS / 64];
    __shared__ uint64_t s_keys[KNZ_RS_TILE];
    __shared__ uint32_t s_vals[KNZ_RS_TILE];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const SsTile t = knz_ss_tile(g, blockIdx.x);
    const uint8_t* src = (const uint8_t*)g.in_ptr[t.b];
    for (uint32_t i = tid; i < (KNZ_RS_THREADS / 64) * 256; i += KNZ_RS_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    uint64_t key[KNZ_RS_ITEMS];
    uint32_t val[KNZ_RS_ITEMS];
    uint32_t rank[KNZ_RS_ITEMS];
#pragma unroll
    for (int r = 0; r < KNZ_RS_ITEMS; r++) {
        const uint32_t loc = t.t0 + w * (64 * KNZ_RS_ITEMS) + (uint32_t)r * 64 + lane;
        const bool valid = loc < t.n;
        if (FIRST) { key[r] = valid ? knz_ss_text_key(src, loc, t.n) : 0; val[r] = t.gs0 + loc; }
        else { key[r] = valid ? kin[t.gs0 + loc] : 0; val[r] = valid ? vin[t.gs0 + loc] : 0u; }
    }
    // rank of a pair among the wave's pairs with the same digit: the lanes of a row that share a digit find each other with one ballot per bit, the first
    // of them adds the row's count to the wave's counter with an LDS atomic whose returned value is the count of the rows in front (LDS atomics of
    // one wave execute in program order): every row is issued before the first result is looked at, no LDS round trip per row
    uint32_t old[KNZ_RS_ITEMS];
#pragma unroll
    for (int r = 0; r < KNZ_RS_ITEMS; r++) {
        const uint32_t loc = t.t0 + w * (64 * KNZ_RS_ITEMS) + (uint32_t)r * 64 + lane;
        const bool valid = loc < t.n;
        const uint32_t d = valid ? (uint32_t)(key[r] >> shift) & 0xFFu : 0u;
        const uint64_t m = knz_match_digit(d, valid);
        const uint32_t below = wave_mbcnt64(m);
        const uint32_t leader = valid ? (uint32_t)__ffsll((unsigned long long)m) - 1 : 0u;
        old[r] = 0;
        if (valid && below == 0) old[r] = atomicAdd(&s_cnt[w][d], (uint32_t)__popcll(m));
        wave_order_lanes();
        rank[r] = (d << 24) | (leader << 8) | below;
    }
#pragma unroll
    for (int r = 0; r < KNZ_RS_ITEMS; r++) {
        const uint32_t before = wave_shfl(old[r], (int)((rank[r] >> 8) & 63u));
        rank[r] = (rank[r] & 0xFF000000u) | (before + (rank[r] & 0xFFu));
    }
    __syncthreads();
    {
        uint32_t c[KNZ_RS_THREADS / 64], tot = 0;
        for (int k = 0; k < KNZ_RS_THREADS / 64; k++) { c[k] = s_cnt[k][tid]; tot += c[k]; }
        uint32_t total;
        const uint32_t first = knz_wg256_scan_incl<KnzOpSum>(tot, s_w, total) - tot;
        uint32_t run = first;
        for (int k = 0; k < KNZ_RS_THREADS / 64; k++) { s_cnt[k][tid] = run; run += c[k]; }
        s_gbase[tid] = offs[(size_t)256 * g.tile_base[t.b] + (size_t)tid * t.tpb + t.lt] - first;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < KNZ_RS_ITEMS; r++) {
        const uint32_t loc = t.t0 + w * (64 * KNZ_RS_ITEMS) + (uint32_t)r * 64 + lane;
        if (loc < t.n) {
            const uint32_t j = s_cnt[w][rank[r] >> 24] + (rank[r] & 0xFFFFFFu);
            s_keys[j] = key[r];
            s_vals[j] = val[r];
        }
    }
    __syncthreads();
    const uint32_t items = t.n - t.t0 < KNZ_RS_TILE ? t.n - t.t0 : KNZ_RS_TILE;
    for (uint32_t j = tid; j < items; j += KNZ_RS_THREADS) {
        const uint64_t kk = s_keys[j];
        const uint32_t pos = s_gbase[(uint32_t)(kk >> shift) & 0xFFu] + j;        // (the flat sum already starts at the block's first slot)
        kout[pos] = kk;
        vout[pos] = s_vals[j];
    }
}

// ---- groups as head bits ---------------------------------------------------------------------------------------------------------------------
// after round 0: bit j = slot j starts a group (a key that differs from its predecessor's, or the first slot of a block)
__global__ __launch_bounds__(256) void knz_ss_heads_kernel(SsGeom g, const uint64_t* keys, uint32_t total, uint64_t* hb) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * 4096u + w * 1024u;
    for (int r = 0; r < 16; r++) {
        const uint32_t j0 = wave_uniform(base + (uint32_t)r * 64);
        if (j0 >= total) break;
        const uint32_t j = j0 + lane;
        const uint64_t key = j < total ? keys[j] : 0ull;
        uint64_t prev = wave_shfl64(key, (int)((lane + 63) & 63));           // the key of the slot in front: the lane below, lane 0 reads it
        if (lane == 0) prev = j ? keys[j - 1] : ~key;
        // a row of 64 slots holds the first slot of a block only if it is its own first slot's block's start, or the next block starts inside it
        const uint32_t b0 = knz_ss_block_of(g.gstart, g.nblocks, j0);
        bool start = j == g.gstart[b0];
        if (g.gstart[b0 + 1] < j0 + 64 && j < total) start = j == g.gstart[knz_ss_block_of(g.gstart, g.nblocks, j)];
        const uint64_t bal = wave_ballot(j < total && (start || key != prev));
        if (lane == 0) hb[j0 >> 6] = bal;
    }
}

// first and last head of every tile of 4096 list entries (one wave per tile): tfirst = index or M, tlast1 = index + 1 or 0
__global__ __launch_bounds__(256) void knz_ss_tile_heads_kernel(const uint64_t* hb, uint32_t M, uint32_t tiles, uint32_t* tfirst, uint32_t* tlast1) {
    const uint32_t lane = threadIdx.x & 63, tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= tiles) return;
    const uint32_t widx = tile * 64 + lane;
    const uint64_t word = (uint64_t)widx * 64 < M ? hb[widx] : 0ull;
    const uint64_t nz = wave_ballot(word != 0);
    if (nz == 0) { if (lane == 0) { tfirst[tile] = M; tlast1[tile] = 0; } return; }
    const int lf = __ffsll((unsigned long long)nz) - 1, ll = 63 - __clzll((long long)nz);
    const uint64_t wf = wave_shfl64(word, lf), wl = wave_shfl64(word, ll);
    if (lane == 0) {
        tfirst[tile] = tile * 4096u + (uint32_t)lf * 64 + (uint32_t)(__ffsll((unsigned long long)wf) - 1);
        tlast1[tile] = tile * 4096u + (uint32_t)ll * 64 + (uint32_t)(63 - __clzll((long long)wl)) + 1;
    }
}

// prev1[t] = 1 + the last head in front of tile t (0: none), nextf[t] = the first head behind tile t (M: none). One workgroup.
__global__ __launch_bounds__(1024) void knz_ss_tile_scan_kernel(const uint32_t* tfirst, const uint32_t* tlast1, uint32_t tiles, uint32_t M, uint32_t* prev1, uint32_t* nextf) {
    __shared__ uint32_t s_w[17];
    __shared__ uint32_t s_x[1024];
    const uint32_t tid = threadIdx.x;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < tiles; base += 1024) {
        const uint32_t i = base + tid;
        const uint32_t v = i < tiles ? tlast1[i] : 0u;
        const uint32_t incl = knz_ss_wg_scan_incl_max<16>(v, s_w);
        s_x[tid] = incl;
        __syncthreads();
        const uint32_t excl = tid ? s_x[tid - 1] : 0u;
        const uint32_t top = s_x[1023];
        __syncthreads();
        if (i < tiles) prev1[i] = carry > excl ? carry : excl;
        carry = carry > top ? carry : top;
    }
    // backwards: min of tfirst over the tiles behind = M - max(M - tfirst)
    carry = 0;
    for (uint32_t base = 0; base < tiles; base += 1024) {
        const uint32_t ri = base + tid;                       // reversed index
        const uint32_t v = ri < tiles ? M - tfirst[tiles - 1 - ri] : 0u;
        const uint32_t incl = knz_ss_wg_scan_incl_max<16>(v, s_w);
        s_x[tid] = incl;
        __syncthreads();
        const uint32_t excl = tid ? s_x[tid - 1] : 0u;
        const uint32_t top = s_x[1023];
        __syncthreads();
        if (ri < tiles) nextf[tiles - 1 - ri] = M - (carry > excl ? carry : excl);
        carry = carry > top ? carry : top;
    }
}

// exclusive sums of up to three arrays of m counters in place; totals to out[0..2]. One workgroup, 16 counters per thread and step.
__global__ __launch_bounds__(1024) void knz_ss_scan_kernel(uint32_t* a0, uint32_t* a1, uint32_t* a2, uint32_t m, uint32_t* out) {
    __shared__ uint32_t s_w[17];
    uint32_t* arr[3] = {a0, a1, a2};
    for (int q = 0; q < 3; q++) {
        uint32_t* a = arr[q];
        if (a == nullptr) continue;
        uint32_t carry = 0;
        for (uint32_t base = 0; base < m; base += 16384) {
            const uint32_t p0 = base + threadIdx.x * 16;
            uint32_t v[16], acc = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) { v[j] = p0 + j < m ? a[p0 + j] : 0u; acc += v[j]; }
            uint32_t total;
            uint32_t run = carry + knz_ss_wg_scan_excl<16>(acc, s_w, total);
#pragma unroll
            for (int j = 0; j < 16; j++) { if (p0 + j < m) a[p0 + j] = run; run += v[j]; }
            carry += total;
        }
        if (threadIdx.x == 0) out[q] = carry;
    }
}

// start (last head at or before) and end (first head behind) of the group of each of a thread's 16 consecutive list entries
struct SsBounds { uint32_t bits, start[16], end[16]; };
__device__ __forceinline__ void knz_ss_bounds(const uint64_t* hb, uint32_t M, uint32_t tile, const uint32_t* prev1, const uint32_t* nextf, uint32_t* s_w, uint32_t* s_x, SsBounds& o) {
    const uint32_t tid = threadIdx.x;
    const uint32_t p0 = tile * 4096u + tid * 16u;
    const uint32_t widx = tile * 64 + (tid >> 2);
    const uint64_t word = (uint64_t)widx * 64 < M ? hb[widx] : 0ull;
    const uint32_t bits = (uint32_t)(word >> ((tid & 3) * 16)) & 0xFFFFu;
    o.bits = bits;
    const uint32_t last1 = bits ? p0 + (31u - (uint32_t)__clz((int)bits)) + 1 : 0u;
    const uint32_t firstInv = bits ? M - (p0 + (uint32_t)(__ffs((int)bits) - 1)) : 0u;      // bigger = closer
    const uint32_t inclL = knz_ss_wg_scan_incl_max<4>(last1, s_w);
    s_x[tid] = inclL;
    __syncthreads();
    uint32_t cs = tid ? s_x[tid - 1] : 0u;
    __syncthreads();
    const uint32_t pv = prev1[tile];
    cs = cs > pv ? cs : pv;                                                                   // 1 + last head in front of this thread's entries
    s_x[255 - tid] = firstInv;
    __syncthreads();
    const uint32_t rv = s_x[tid];
    __syncthreads();
    const uint32_t inclF = knz_ss_wg_scan_incl_max<4>(rv, s_w);
    s_x[tid] = inclF;
    __syncthreads();
    const uint32_t rtid = 255 - tid;
    uint32_t ce = rtid ? s_x[rtid - 1] : 0u;
    __syncthreads();
    uint32_t nx = nextf[tile];
    nx = nx > M ? M : nx;
    uint32_t cend = M - ce;                                                                   // first head behind this thread's entries
    cend = cend < nx ? cend : nx;
    uint32_t cur = cs ? cs - 1 : 0u;
#pragma unroll
    for (int q = 0; q < 16; q++) { if (bits & (1u << q)) cur = p0 + q; o.start[q] = cur; }
    uint32_t ne = cend;
#pragma unroll
    for (int q = 15; q >= 0; q--) { o.end[q] = ne; if (bits & (1u << q)) ne = p0 + q; }
}

struct SsClassArgs {
    SsGeom g;
    const uint64_t* hb; uint32_t M;                   // the list (M entries) and its head bits
    const uint32_t* lpos;                             // list mode: entry -> slot | HEAD(old) ; identity mode: null (entry = slot)
    const uint32_t* sa; uint32_t* rank;
    const uint32_t* prev1; const uint32_t* nextf;
    uint32_t* cnt_n; uint32_t* cnt_l; uint32_t* cnt_lh;   // [tiles] counts, then exclusive sums
    const uint32_t* base_n;                           // device word: entries the normal list holds already
    uint32_t* pos_n; uint32_t* pos_l; uint32_t* gid_l;
    uint32_t T;                                       // largest group of the normal list
};

// entries of groups of 2..T / more than T members / heads among the latter, per tile
__global__ __launch_bounds__(256) void knz_ss_class_count_kernel(SsClassArgs a) {
    __shared__ uint32_t s_w[5];
    __shared__ uint32_t s_x[256];
    SsBounds bd;
    knz_ss_bounds(a.hb, a.M, blockIdx.x, a.prev1, a.nextf, s_w, s_x, bd);
    const uint32_t p0 = blockIdx.x * 4096u + threadIdx.x * 16u;
    uint32_t nn = 0, nl = 0, nlh = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        if (p0 + q >= a.M) break;
        const uint32_t size = bd.end[q] - bd.start[q];
        if (size > a.T) { nl++; if (bd.start[q] == p0 + q) nlh++; }
        else if (size > 1) nn++;
    }
    uint32_t t0, t1, t2;
    knz_ss_wg_scan_excl<4>(nn, s_w, t0);
    knz_ss_wg_scan_excl<4>(nl, s_w, t1);
    knz_ss_wg_scan_excl<4>(nlh, s_w, t2);
    if (threadIdx.x == 0) { a.cnt_n[blockIdx.x] = t0; a.cnt_l[blockIdx.x] = t1; a.cnt_lh[blockIdx.x] = t2; }
}

// ranks of every entry; the two lists of the next round
template <bool IDENT>
__global__ __launch_bounds__(256) void knz_ss_class_write_kernel(SsClassArgs a) {
    __shared__ uint32_t s_w[5];
    __shared__ uint32_t s_x[256];
    __shared__ uint32_t s_sa[4096];                    // the tile's suffixes and (list mode) entries, loaded with consecutive threads on consecutive words;
    __shared__ uint32_t s_j[IDENT ? 1 : 4096];         // a thread then owns 16 CONSECUTIVE entries (the carries of the group bounds run along them)
    __shared__ uint32_t s_n[4096];                     // the tile's entries for the normal list, written out the same way
    const uint32_t t0 = blockIdx.x * 4096u;
    for (uint32_t x = threadIdx.x; x < 4096; x += 256) {
        const uint32_t k = t0 + x;
        if (k < a.M) {
            if (IDENT) s_sa[x] = a.sa[k];
            else { const uint32_t e = a.lpos[k]; s_j[x] = e; s_sa[x] = a.sa[e & KNZ_SS_MASK]; }
        }
    }
    SsBounds bd;
    knz_ss_bounds(a.hb, a.M, blockIdx.x, a.prev1, a.nextf, s_w, s_x, bd);     // (barriers inside: the staged words are visible behind it)
    const uint32_t x0 = threadIdx.x * 16u, p0 = t0 + x0;
    uint32_t nn = 0, nl = 0, nlh = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
        if (p0 + q >= a.M) break;
        const uint32_t size = bd.end[q] - bd.start[q];
        if (size > a.T) { nl++; if (bd.start[q] == p0 + q) nlh++; }
        else if (size > 1) nn++;
    }
    uint32_t totN, tt;
    uint32_t on = knz_ss_wg_scan_excl<4>(nn, s_w, totN);
    uint32_t ol = knz_ss_wg_scan_excl<4>(nl, s_w, tt) + a.cnt_l[blockIdx.x];
    uint32_t olh = knz_ss_wg_scan_excl<4>(nlh, s_w, tt) + a.cnt_lh[blockIdx.x];
    // the first slot of a block, for the block-local ranks: all of a tile's group starts lie in one block almost always
    const uint32_t bFirst = knz_ss_block_of(a.g.gstart, a.g.nblocks, IDENT ? (a.prev1[blockIdx.x] ? a.prev1[blockIdx.x] - 1 : 0u) : 0u);
    const uint32_t gsF = a.g.gstart[bFirst], geF = a.g.gstart[bFirst + 1];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const uint32_t k = p0 + q;
        if (k >= a.M) break;
        const uint32_t st = bd.start[q], size = bd.end[q] - st;
        uint32_t j, js;
        bool same = false;
        if (IDENT) { j = k; js = st; }
        else {
            j = s_j[x0 + q] & KNZ_SS_MASK;
            const uint32_t e = st >= t0 ? s_j[st - t0] : a.lpos[st];
            js = e & KNZ_SS_MASK; same = (e & KNZ_SS_HEAD) != 0;
        }
        const bool head = st == k;
        if (!same) {
            const uint32_t gs = (IDENT && js >= gsF && js < geF) ? gsF : a.g.gstart[knz_ss_block_of(a.g.gstart, a.g.nblocks, js)];
            a.rank[s_sa[x0 + q]] = js - gs + 1;
        }
        if (size > a.T) {
            if (head) olh++;
            a.pos_l[ol] = j | (head ? KNZ_SS_HEAD : 0u);
            a.gid_l[ol] = olh - 1;
            ol++;
        } else if (size > 1) {
            s_n[on++] = j | (head ? KNZ_SS_HEAD : 0u);
        }
    }
    __syncthreads();
    const uint32_t base = a.cnt_n[blockIdx.x] + *a.base_n;
    for (uint32_t x = threadIdx.x; x < totN; x += 256) a.pos_n[base + x] = s_n[x];
}

// ---- normal list: segmented sort of one round ------------------------------------------------------------------------------------------------
#ifdef KNZ_SG_PROF
#define KNZ_SG_TICK(i) do { if (threadIdx.x == 0) { const long long t__ = clock64(); atomicAdd((unsigned long long*)&a.prof[i], (unsigned long long)(t__ - tprev)); tprev = t__; } } while (0)
#else
#define KNZ_SG_TICK(i) do { } while (0)
#endif
struct SgArgs {
    unsigned long long* prof;             // (KNZ_SG_PROF builds: cycles per phase, summed over the workgroups)
    SsGeom g;
    uint32_t m;                           // entries of the list
    const uint32_t* pos;                  // [m] slot | HEAD
    uint32_t* sa;                         // refined in place
    uint32_t* rank;
    const uint32_t* key2; const uint32_t* suf;   // [m] second key and suffix of every entry (knz_sg_keys_kernel)
    uint32_t* res;                        // [m] new group start (slot) | SINGLE | SAME, by list index
    uint32_t* tile_a;                     // [tiles] first list index of the stretch the workgroup took
    uint32_t* tile_cnt;                   // [tiles] entries that stay unresolved
    uint32_t* pos_out;                    // update: the list of the next round
    const uint32_t* tile_off;             // update: exclusive sums of tile_cnt
    uint32_t h, rbits;
    int32_t* err;
};

template <int T, int THREADS>
__global__ __launch_bounds__(THREADS) void knz_sg_sort_kernel(SgArgs a) {
    constexpr int CAP = 2 * T, NW = THREADS / 64, ROWS = CAP / THREADS, DPT = 512 / THREADS;
    __shared__ uint64_t s_keys[CAP];
    __shared__ uint32_t s_vals[CAP];
    __shared__ uint32_t s_cnt[NW][512];
    __shared__ uint64_t s_hb[CAP / 64 + 1];
    __shared__ uint32_t s_w[NW + 1];
    __shared__ uint32_t s_a0, s_a1, s_surv;
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, tile = blockIdx.x;
    const uint32_t m = a.m, k0 = tile * (uint32_t)T;
    const uint64_t le = lane == 63 ? ~0ull : ((2ull << lane) - 1);
    if (tid == 0) { s_a0 = CAP; s_a1 = CAP; s_surv = 0; }
#ifdef KNZ_SG_PROF
    long long tprev = clock64();
#endif
    __syncthreads();
    {   // the stretch of the list this workgroup takes = [first head at or behind k0, first head at or behind k0 + T): found in a window of 2T entries
        uint32_t v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) { const uint32_t k = k0 + (uint32_t)r * THREADS + tid; v[r] = k < m ? a.pos[k] : KNZ_SS_HEAD; }
        uint32_t mn0 = CAP, mn1 = CAP;
#pragma unroll
        for (int r = ROWS - 1; r >= 0; r--) {
            if (v[r] & KNZ_SS_HEAD) { const uint32_t x = (uint32_t)r * THREADS + tid; if (x < (uint32_t)T) mn0 = x; else mn1 = x; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o0 = wave_shfl(mn0, (int)(lane ^ (uint32_t)d)), o1 = wave_shfl(mn1, (int)(lane ^ (uint32_t)d));
            mn0 = mn0 < o0 ? mn0 : o0; mn1 = mn1 < o1 ? mn1 : o1;
        }
        if (lane == 0) { atomicMin(&s_a0, mn0); atomicMin(&s_a1, mn1); }
    }
    __syncthreads();
    uint32_t a0 = s_a0, a1 = s_a1;
    if (a0 >= (uint32_t)T || a1 >= (uint32_t)CAP) {                                   // a group of more than T entries on this list: must not happen
        if (tid == 0) { *a.err = 1; a.tile_a[tile] = k0; a.tile_cnt[tile] = 0; }
        return;
    }
    const uint32_t left = m - k0;
    a0 = a0 < left ? a0 : left; a1 = a1 < left ? a1 : left;
    const uint32_t n = a1 - a0, kb = k0 + a0;
    if (n == 0) { if (tid == 0) { a.tile_a[tile] = kb; a.tile_cnt[tile] = 0; } return; }
    const uint32_t rows = (n + THREADS - 1) / THREADS, eb = w * rows * 64;
    const uint32_t rbits = a.rbits;
    uint32_t pj[ROWS];                                                                // list entries of this thread's positions (slot | HEAD): the positions stay, the suffixes move
    uint64_t key[ROWS];
    uint32_t val[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const uint32_t e = eb + (uint32_t)r * 64 + lane;
        pj[r] = ((uint32_t)r < rows && e < n) ? a.pos[kb + e] : 0u;
    }
#pragma unroll
    for (int r = 0; r < ROWS; r++) {                                                  // second keys and suffixes: gathered by knz_sg_keys_kernel (a streaming kernel: every
        const uint32_t e = eb + (uint32_t)r * 64 + lane;                              // lane of the chip has a gather in flight there; here they would wait between barriers)
        const bool valid = (uint32_t)r < rows && e < n;
        key[r] = valid ? (uint64_t)a.key2[kb + e] : 0ull;
        val[r] = valid ? a.suf[kb + e] : 0u;
    }
    KNZ_SG_TICK(0);
    uint32_t hcount = 0;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        if ((uint32_t)r < rows) {
            const uint64_t bal = wave_ballot((pj[r] & KNZ_SS_HEAD) != 0);                // (positions behind n hold 0)
            key[r] |= (uint64_t)(hcount + (uint32_t)__popcll(bal & le)) << rbits;        // 1-based among the wave's heads
            hcount += (uint32_t)__popcll(bal);
        }
    }
    if (lane == 0) s_w[w] = hcount;
    __syncthreads();
    KNZ_SG_TICK(1);
    uint32_t woff = 0, ng = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { const uint32_t x = s_w[k]; if ((uint32_t)k < w) woff += x; ng += x; }
#pragma unroll
    for (int r = 0; r < ROWS; r++) key[r] += (uint64_t)woff << rbits;                     // group number 1..ng in front of the second key
    const uint32_t gbits = 32u - (uint32_t)__clz((int)ng);                               // values 1..ng
    const uint32_t nbits = rbits + gbits;
    const uint32_t passes = (nbits + 8) / 9, dbits = (nbits + passes - 1) / passes;
    const uint32_t dmask = (1u << dbits) - 1;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t shift = p * dbits;
        __syncthreads();                                                                  // (s_w, and the counters of the pass before, have been read)
        for (uint32_t i = tid; i < (uint32_t)NW * 512; i += THREADS) (&s_cnt[0][0])[i] = 0;
        __syncthreads();
        // rank of every item among the wave's items with the same digit: lanes of a row that share a digit find each other with one ballot per bit;
        // the row's first such lane adds the row's count to the wave's counter of the digit (LDS atomics of one wave execute in program order, so
        // the returned values ARE the counts of the rows in front: all rows are issued before the first result is looked at)
        uint32_t rk[ROWS], old[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            rk[r] = 0; old[r] = 0;
            if ((uint32_t)r < rows) {
                const uint32_t e = eb + (uint32_t)r * 64 + lane;
                const bool valid = e < n;
                const uint32_t d = valid ? (uint32_t)(key[r] >> shift) & dmask : 0u;
                const uint64_t mm = knz_ss_match(d, valid, (int)dbits);
                const uint32_t below = wave_mbcnt64(mm);
                const uint32_t leader = valid ? (uint32_t)__ffsll((unsigned long long)mm) - 1 : 0u;
                if (valid && below == 0) old[r] = atomicAdd(&s_cnt[w][d], (uint32_t)__popcll(mm));
                wave_order_lanes();
                rk[r] = (d << 16) | (leader << 8) | below;
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            if ((uint32_t)r < rows) {
                const uint32_t before = wave_shfl(old[r], (int)((rk[r] >> 8) & 63u));
                rk[r] = (rk[r] & 0xFFFF0000u) | (before + (rk[r] & 0xFFu));
            }
        }
        __syncthreads();
        {
            uint32_t c[DPT][NW], tot = 0;
#pragma unroll
            for (int q = 0; q < DPT; q++)
#pragma unroll
                for (int k = 0; k < NW; k++) { c[q][k] = s_cnt[k][tid * DPT + q]; tot += c[q][k]; }
            uint32_t total;
            uint32_t run = knz_ss_wg_scan_excl<NW>(tot, s_w, total);
#pragma unroll
            for (int q = 0; q < DPT; q++)
#pragma unroll
                for (int k = 0; k < NW; k++) { s_cnt[k][tid * DPT + q] = run; run += c[q][k]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const uint32_t e = eb + (uint32_t)r * 64 + lane;
            if ((uint32_t)r < rows && e < n) { const uint32_t dst = s_cnt[w][rk[r] >> 16] + (rk[r] & 0xFFFFu); s_keys[dst] = key[r]; s_vals[dst] = val[r]; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const uint32_t e = eb + (uint32_t)r * 64 + lane;
            if ((uint32_t)r < rows && e < n) { key[r] = s_keys[e]; val[r] = s_vals[e]; }
        }
    }
    KNZ_SG_TICK(2);
    // new groups: a key that differs from its predecessor's
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        if ((uint32_t)r < rows) {
            const uint32_t e = eb + (uint32_t)r * 64 + lane;
            const bool hd = e < n && (e == 0 || key[r] != s_keys[e - 1]);
            const uint64_t bal = wave_ballot(hd);
            if (lane == 0) s_hb[w * rows + (uint32_t)r] = bal;
        }
    }
    if (tid == 0) s_hb[NW * rows] = ~0ull;
    __syncthreads();
    KNZ_SG_TICK(3);
    uint32_t surv = 0;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const uint32_t e = eb + (uint32_t)r * 64 + lane;
        if ((uint32_t)r < rows && e < n) {
            uint32_t word = e >> 6;
            uint64_t msk = s_hb[word] & (((e & 63) == 63) ? ~0ull : ((2ull << (e & 63)) - 1));
            const bool hd = (msk >> (e & 63)) & 1ull;
            while (msk == 0) msk = s_hb[--word];
            const uint32_t ps = word * 64 + (63u - (uint32_t)__clzll((long long)msk));
            const bool nexth = e + 1 == n || ((s_hb[(e + 1) >> 6] >> ((e + 1) & 63)) & 1ull);
            const bool single = hd && nexth;
            const uint32_t jst = a.pos[kb + ps];                                          // (the workgroup has just read this stretch: L1 / L2)
            a.sa[pj[r] & KNZ_SS_MASK] = val[r];
            a.res[kb + e] = (jst & KNZ_SS_MASK) | (single ? KNZ_SS_SINGLE : 0u) | ((jst & KNZ_SS_HEAD) ? KNZ_SS_SAME : 0u);
            if (!single) surv++;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) surv += wave_shfl(surv, (int)(lane ^ (uint32_t)d));
    if (lane == 0) atomicAdd(&s_surv, surv);
    __syncthreads();
    KNZ_SG_TICK(4);
#ifdef KNZ_SG_PROF
    if (tid == 0) { atomicAdd(&a.prof[5], (unsigned long long)passes); atomicAdd(&a.prof[6], (unsigned long long)n); atomicAdd(&a.prof[7], 1ull); atomicAdd(&a.prof[8], (unsigned long long)ng); }
#endif
    if (tid == 0) { a.tile_a[tile] = kb; a.tile_cnt[tile] = s_surv; }
}

// second key of every list entry: rank of the suffix h symbols on (lifted over h), or length - 1 for a suffix that ends before; and the suffix itself
__global__ __launch_bounds__(256) void knz_sg_keys_kernel(SsGeom g, const uint32_t* pos, uint32_t m, const uint32_t* sa, const uint32_t* rank, uint32_t h,
                                                          uint32_t* key2, uint32_t* suf) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    const uint32_t j = pos[k] & KNZ_SS_MASK;
    const uint32_t i = sa[j];
    const uint32_t end = g.gstart[knz_ss_block_of(g.gstart, g.nblocks, j) + 1];
    const uint64_t i2 = (uint64_t)i + h;
    key2[k] = i2 < end ? rank[i2] + h : end - 1 - i;
    suf[k] = i;
}

// ranks of the refined stretch, and the entries that stay on the list (order kept: groups stay contiguous)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void knz_sg_update_kernel(SgArgs a, uint32_t tiles) {
    constexpr int NW = THREADS / 64;
    __shared__ uint32_t s_w[NW + 1];
    __shared__ uint32_t s_b0, s_b1;
    const uint32_t tid = threadIdx.x, tile = blockIdx.x;
    const uint32_t A = a.tile_a[tile], B = tile + 1 < tiles ? a.tile_a[tile + 1] : a.m;
    if (A >= B) return;
    if (tid == 0) { s_b0 = 0xFFFFFFFFu; s_b1 = 0; }
    __syncthreads();
    {   // blocks the stretch touches (the list is not in slot order once entries came over from the large list)
        uint32_t jmin = 0xFFFFFFFFu, jmax = 0;
        for (uint32_t k = A + tid; k < B; k += THREADS) { const uint32_t j = a.pos[k] & KNZ_SS_MASK; jmin = jmin < j ? jmin : j; jmax = jmax > j ? jmax : j; }
        const uint32_t lane = tid & 63;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t o0 = wave_shfl(jmin, (int)(lane ^ (uint32_t)d)), o1 = wave_shfl(jmax, (int)(lane ^ (uint32_t)d));
            jmin = jmin < o0 ? jmin : o0; jmax = jmax > o1 ? jmax : o1;
        }
        if (lane == 0) { atomicMin(&s_b0, jmin); atomicMax(&s_b1, jmax); }
    }
    __syncthreads();
    const bool oneBlock = knz_ss_block_of(a.g.gstart, a.g.nblocks, s_b0) == knz_ss_block_of(a.g.gstart, a.g.nblocks, s_b1);
    const uint32_t gsU = a.g.gstart[knz_ss_block_of(a.g.gstart, a.g.nblocks, s_b0)];
    uint32_t off = a.tile_off[tile];
    for (uint32_t base = A; base < B; base += THREADS) {
        const uint32_t k = base + tid;
        bool surv = false;
        uint32_t j = 0, js = 0;
        if (k < B) {
            const uint32_t v = a.res[k];
            j = a.pos[k] & KNZ_SS_MASK; js = v & KNZ_SS_MASK;
            if (!(v & KNZ_SS_SAME)) a.rank[a.sa[j]] = js - (oneBlock ? gsU : a.g.gstart[knz_ss_block_of(a.g.gstart, a.g.nblocks, js)]) + 1;
            surv = !(v & KNZ_SS_SINGLE);
        }
        uint32_t total;
        const uint32_t idx = knz_ss_wg_scan_excl<NW>(surv ? 1u : 0u, s_w, total);
        if (surv) a.pos_out[off + idx] = j | (js == j ? KNZ_SS_HEAD : 0u);
        off += total;
    }
}

// ---- large list: keys for the device-wide sort, results back into the suffix array ------------------------------------------------------------
// Long runs of one symbol (zero pages, padding) are what keeps groups large for many rounds: doubling peels [h, 2h) symbols off the end of a run per
// round. Once the large list is short its keys are made RUN-AWARE instead: a suffix inside a run of c with L >= h symbols of the run left (so its whole
// group is c^h...) is ordered among its group by what follows the run: t = the symbol behind the run (the end of the block counts as smaller than
// every symbol). t < c: the suffix sorts in front of every member with a longer rest of the run (class 0, L ascending); t > c: behind them (class 1,
// L descending); equal class and L: by the rank of the suffix that starts behind the run. One round then resolves a run whatever its length.
// Run ends come from a bit per text position ("differs from the position in front", knz_ss_run_bits_kernel) and the per-tile "first bit behind
// this tile" table the group bounds use.
// A thread takes 8 positions (one 8-byte read + the byte in front) and stores their 8 bits as one byte of the little-endian bit map (the first
// form took a position per lane and a ballot per row: 0.64 ms for 212 MB).
__global__ __launch_bounds__(256) void knz_ss_run_bits_kernel(SsGeom g, uint32_t total, uint64_t* rb) {
    uint8_t* rb8 = (uint8_t*)rb;
    const uint32_t words8 = ((total + 63u) >> 6) << 3;                            // bytes of the map (whole 64-bit words)
    for (uint32_t r = 0; r < 2u; r++) {
        const uint32_t i = blockIdx.x * 4096u + r * 2048u + threadIdx.x * 8u;
        if ((i >> 3) >= words8) break;
        uint32_t bits = 0;
        if (i < total) {
            const uint32_t w0 = wave_uniform(i - (threadIdx.x & 63u) * 8u);       // the wave's 512 positions
            uint32_t b = knz_ss_block_of(g.gstart, g.nblocks, w0);
            if (g.gstart[b + 1] < w0 + 512u) b = knz_ss_block_of(g.gstart, g.nblocks, i);
            const uint32_t bend = g.gstart[b + 1];
            if (i + 8u <= bend) {                                                   // all eight inside block b
                const uint8_t* src = (const uint8_t*)g.in_ptr[b];
                const uint32_t loc = i - g.gstart[b];
                const uint64_t x = knz_vle64(src + loc);
                const uint64_t d = x ^ ((x << 8) | (uint64_t)(loc ? src[loc - 1] : (uint8_t)~(uint8_t)x));
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) bits |= ((d >> (8u * k)) & 0xFFu) ? 1u << k : 0u;
            } else {
                for (uint32_t k = 0; k < 8u && i + k < total; k++) {
                    const uint32_t bb = knz_ss_block_of(g.gstart, g.nblocks, i + k);
                    const uint8_t* src = (const uint8_t*)g.in_ptr[bb];
                    const uint32_t loc = i + k - g.gstart[bb];
                    if (loc == 0 || src[loc] != src[loc - 1]) bits |= 1u << k;
                }
            }
        }
        rb8[i >> 3] = (uint8_t)bits;
    }
}

// first position behind i whose bit is set (total: none)
__device__ __forceinline__ uint32_t knz_ss_run_end(const uint64_t* rb, const uint32_t* rnextf, uint32_t total, uint32_t i) {
    const uint32_t p = i + 1;
    if (p >= total) return total;
    uint32_t w = p >> 6;
    uint64_t bits = rb[w] >> (p & 63);
    if (bits) return p + (uint32_t)(__ffsll((unsigned long long)bits) - 1);
    const uint32_t wend = (w | 63u);                                            // last word of the tile of 4096 positions
    const uint32_t wlast = (total - 1) >> 6;
    for (w++; w <= wend && w <= wlast; w++) {
        bits = rb[w];
        if (bits) return w * 64 + (uint32_t)(__ffsll((unsigned long long)bits) - 1);
    }
    const uint32_t nx = rnextf[p >> 12];
    return nx < total ? nx : total;
}

struct SlKeyArgs {
    SsGeom g;
    const uint32_t* lpos; const uint32_t* gid; uint32_t m;
    const uint32_t* sa; const uint32_t* rank;
    uint32_t h, pbits;                     // bits of the field behind the group number
    const uint64_t* rb; const uint32_t* rnextf; uint32_t total;     // run-aware keys (rb == null: plain doubling keys)
    uint32_t lmax, lbits, rkbits;
    uint64_t* keys; uint32_t* vals;
};

__global__ __launch_bounds__(256) void knz_sl_keys_kernel(SlKeyArgs a) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= a.m) return;
    const uint32_t j = a.lpos[k] & KNZ_SS_MASK;
    const uint32_t i = a.sa[j];
    const uint32_t b = knz_ss_block_of(a.g.gstart, a.g.nblocks, j);
    const uint32_t gs0 = a.g.gstart[b], end = a.g.gstart[b + 1];
    uint64_t field;
    bool run = false;
    if (a.rb) {
        uint32_t e = knz_ss_run_end(a.rb, a.rnextf, a.total, i);
        e = e < end ? e : end;
        const uint32_t L = e - i;
        if (L >= a.h) {
            run = true;
            const uint8_t* src = (const uint8_t*)a.g.in_ptr[b];
            const bool up = e < end && src[e - gs0] > src[i - gs0];              // the run ends on a larger symbol: class 1 (the end of the block: class 0)
            const uint64_t rk = e < end ? a.rank[e] : 0u;
            field = ((uint64_t)(up ? 1u : 0u) << (a.lbits + a.rkbits)) | ((uint64_t)(up ? a.lmax - L : L) << a.rkbits) | rk;
        }
    }
    if (!run) {
        const uint64_t i2 = (uint64_t)i + a.h;
        field = i2 < end ? (uint64_t)a.rank[i2] + a.h : (uint64_t)(end - 1 - i);
    }
    a.keys[k] = ((uint64_t)a.gid[k] << a.pbits) | field;
    a.vals[k] = i;
}

__global__ __launch_bounds__(256) void knz_sl_writeback_kernel(const uint32_t* lpos, uint32_t m, const uint64_t* skeys, const uint32_t* svals, uint32_t* sa, uint64_t* hb) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t base = blockIdx.x * 4096u + w * 1024u;
    for (int r = 0; r < 16; r++) {
        const uint32_t k0 = base + (uint32_t)r * 64;
        if (k0 >= m) break;
        const uint32_t k = k0 + lane;
        bool head = false;
        if (k < m) {
            sa[lpos[k] & KNZ_SS_MASK] = svals[k];
            head = k == 0 || skeys[k] != skeys[k - 1];
        }
        const uint64_t bal = wave_ballot(head);
        if (lane == 0) hb[k0 >> 6] = bal;
    }
}
