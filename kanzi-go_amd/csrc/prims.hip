// Device-wide primitives of the suffix sort and the inverse BWT, hand-written for gfx950: LSD radix sort of (key, value) pairs, inclusive
// max-scan, selection of flagged indexes. (Round 1 used rocPRIM's.)
//
// Radix sort: 8-bit digits, least significant first, stable. One pass = three launches over tiles of 4096 pairs (256 threads x 16):
//   hist     per-tile digit counts, stored digit-major (hist[digit * tiles + tile]) so that
//   scan     one flat exclusive prefix sum over the 256 x tiles counters gives every (digit, tile) its first output slot,
//   scatter  each wave ranks its 1024 pairs 64 at a time: lanes that hold the same digit find each other with 8 ballots (match-any), their
//            order inside the group is the popcount of the lower lanes, the wave's running count of that digit sits in LDS; wave totals and
//            digit totals are prefix-summed, the tile is laid out in sorted order in LDS (48 KB) and written out by consecutive threads to
//            consecutive slots (one run of slots per digit and tile).
// HBM traffic per pass: keys read twice, values once, both written once = 32 B per (u64, u32) pair. The passes ping-pong between the caller's
// two buffer pairs (the input pair is scratch), a third pair from `tmp` is used for one hop when the pass count is even so that the result
// always lands in (kout, vout).

#define KNZ_RS_THREADS 256
#define KNZ_RS_ITEMS 16
#define KNZ_RS_TILE (KNZ_RS_THREADS * KNZ_RS_ITEMS)

struct KnzOpSum { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a + b; } };
struct KnzOpMax { static __device__ __forceinline__ uint32_t f(uint32_t a, uint32_t b) { return a > b ? a : b; } };

template <typename OP>
__device__ __forceinline__ uint32_t knz_wave_scan_incl_op(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = wave_shfl(v, (int)((lane - (uint32_t)d) & 63u)); if ((int)lane >= d) v = OP::f(v, o); }
    return v;
}
// inclusive scan of one value per thread over a 256-thread workgroup; total = the reduction of all
template <typename OP>
__device__ __forceinline__ uint32_t knz_wg256_scan_incl(uint32_t v, uint32_t* s_w, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = knz_wave_scan_incl_op<OP>(v, lane);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t acc = 0, tot = 0;
    bool have = false;
    for (uint32_t k = 0; k < KNZ_RS_THREADS / 64; k++) {
        const uint32_t x = s_w[k];
        if (k < w) { acc = have ? OP::f(acc, x) : x; have = true; }
        tot = k ? OP::f(tot, x) : x;
    }
    __syncthreads();
    total = tot;
    return have ? OP::f(acc, incl) : incl;
}


// lanes of the wave whose 8-bit digit equals this lane's (inactive lanes pass d = 256 + something unique-ish and are ignored by the caller)
__device__ __forceinline__ uint64_t knz_match_digit(uint32_t d, bool valid) {
    uint64_t m = wave_ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
        const bool one = (d >> bit) & 1u;
        const uint64_t b = wave_ballot(one);
        m &= one ? b : ~b;
    }
    return m;
}

template <typename K>
__global__ __launch_bounds__(KNZ_RS_THREADS) void knz_rs_hist_kernel(const K* keys, uint32_t n, unsigned shift, uint32_t mask, uint32_t* hist, uint32_t tiles) {
    // counts only: plain LDS atomics into four copies per wave (round 4; the matching of the scatter made this kernel compute-bound)
    __shared__ uint32_t s_cnt[KNZ_RS_THREADS / 64][4][256];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, tile = blockIdx.x;
    for (uint32_t i = tid; i < (KNZ_RS_THREADS / 64) * 4 * 256; i += KNZ_RS_THREADS) (&s_cnt[0][0][0])[i] = 0;
    __syncthreads();
    uint32_t* mine = s_cnt[w][lane & 3];
    const uint64_t base = (uint64_t)tile * KNZ_RS_TILE + (uint64_t)w * (64 * KNZ_RS_ITEMS);
#pragma unroll
    for (int r = 0; r < KNZ_RS_ITEMS; r++) {
        const uint64_t idx = base + (uint64_t)r * 64 + lane;
        if (idx < n) atomicAdd(&mine[(uint32_t)(keys[idx] >> shift) & mask], 1u);
    }
    __syncthreads();
    uint32_t sum = 0;
    for (int k = 0; k < KNZ_RS_THREADS / 64; k++) sum += s_cnt[k][0][tid] + s_cnt[k][1][tid] + s_cnt[k][2][tid] + s_cnt[k][3][tid];
    hist[(size_t)tid * tiles + tile] = sum;
}

template <typename K>
__global__ __launch_bounds__(KNZ_RS_THREADS) void knz_rs_scatter_kernel(const K* kin, const uint32_t* vin, K* kout, uint32_t* vout, uint32_t n, unsigned shift,
                                                                      uint32_t mask, const uint32_t* offs, uint32_t tiles) {
    __shared__ uint32_t s_cnt[KNZ_RS_THREADS / 64][256];
    __shared__ uint32_t s_gbase[256];                                        // first output slot of the digit minus its first slot inside the tile
    __shared__ uint32_t s_w[KNZ_RS_THREADS / 64];
    __shared__ K s_keys[KNZ_RS_TILE];                                        // the tile in sorted order: written out with consecutive threads on consecutive slots
    __shared__ uint32_t s_vals[KNZ_RS_TILE];
    const uint32_t tid = threadIdx.x, lane = tid & 63, w = tid >> 6, tile = blockIdx.x;
    for (uint32_t i = tid; i < (KNZ_RS_THREADS / 64) * 256; i += KNZ_RS_THREADS) (&s_cnt[0][0])[i] = 0;
    __syncthreads();
    const uint64_t tbase = (uint64_t)tile * KNZ_RS_TILE;
    const uint64_t base = tbase + (uint64_t)w * (64 * KNZ_RS_ITEMS);
    K key[KNZ_RS_ITEMS];
    uint32_t val[KNZ_RS_ITEMS];
    uint32_t rank[KNZ_RS_ITEMS];                                              // digit in the top byte, rank inside the wave below
#pragma unroll
    for (int r = 0; r < KNZ_RS_ITEMS; r++) {
        const uint64_t idx = base + (uint64_t)r * 64 + lane;
        const bool valid = idx < n;
        key[r] = valid ? kin[idx] : (K)0;
        val[r] = valid ? vin[idx] : 0u;
    }
    // (round 4) the first lane of every group of equal digits adds the row's count with a returned LDS atomic; the results are looked at behind the last
    // row: LDS atomics of one wave execute in program order, so the returned values are the counts of the rows in front
    uint32_t old[KNZ_RS_ITEMS];
#pragma unroll
    for (int r = 0; r < KNZ_RS_ITEMS; r++) {
        const uint64_t idx = base + (uint64_t)r * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = valid ? (uint32_t)(key[r] >> shift) & mask : 0u;
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
    {   // digit `tid`: its pairs start at `first` inside the sorted tile, each wave's share behind the waves in front of it
        uint32_t c[KNZ_RS_THREADS / 64], tot = 0;
        for (int k = 0; k < KNZ_RS_THREADS / 64; k++) { c[k] = s_cnt[k][tid]; tot += c[k]; }
        uint32_t total;
        const uint32_t first = knz_wg256_scan_incl<KnzOpSum>(tot, s_w, total) - tot;
        uint32_t run = first;
        for (int k = 0; k < KNZ_RS_THREADS / 64; k++) { s_cnt[k][tid] = run; run += c[k]; }
        s_gbase[tid] = offs[(size_t)tid * tiles + tile] - first;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < KNZ_RS_ITEMS; r++) {
        const uint64_t idx = base + (uint64_t)r * 64 + lane;
        if (idx < n) {
            const uint32_t j = s_cnt[w][rank[r] >> 24] + (rank[r] & 0xFFFFFFu);
            s_keys[j] = key[r];
            s_vals[j] = val[r];
        }
    }
    __syncthreads();
    const uint32_t items = (uint32_t)(((uint64_t)n - tbase) < KNZ_RS_TILE ? ((uint64_t)n - tbase) : KNZ_RS_TILE);
    for (uint32_t j = tid; j < items; j += KNZ_RS_THREADS) {
        const K kk = s_keys[j];
        const uint32_t pos = s_gbase[(uint32_t)(kk >> shift) & mask] + j;
        kout[pos] = kk;
        vout[pos] = s_vals[j];
    }
}

// ---- flat scans over u32 arrays: tiles of 4096 (16 consecutive elements per thread), tile totals scanned by one workgroup ------------------
// per-tile reduction: sums[tile] = OP over in[tile * 4096 ..)
template <typename OP>
__global__ __launch_bounds__(KNZ_RS_THREADS) void knz_scan_reduce_kernel(const uint32_t* in, uint64_t n, uint32_t* sums) {
    __shared__ uint32_t s_w[KNZ_RS_THREADS / 64];
    const uint64_t p0 = (uint64_t)blockIdx.x * KNZ_RS_TILE + (uint64_t)threadIdx.x * KNZ_RS_ITEMS;
    uint32_t acc = 0;
    for (int j = 0; j < KNZ_RS_ITEMS; j++) if (p0 + j < n) acc = OP::f(acc, in[p0 + j]);
    uint32_t total;
    knz_wg256_scan_incl<OP>(acc, s_w, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
// sums[0..m) -> exclusive scan in place (identity 0: all values are unsigned), by one workgroup
template <typename OP>
__global__ __launch_bounds__(KNZ_RS_THREADS) void knz_scan_sums_kernel(uint32_t* sums, uint32_t m) {
    __shared__ uint32_t s_w[KNZ_RS_THREADS / 64];
    __shared__ uint32_t s_prev[KNZ_RS_THREADS];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < m; base += KNZ_RS_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < m ? sums[i] : 0u;
        uint32_t total;
        const uint32_t incl = knz_wg256_scan_incl<OP>(v, s_w, total);
        s_prev[threadIdx.x] = incl;
        __syncthreads();
        const uint32_t excl = threadIdx.x ? s_prev[threadIdx.x - 1] : 0u;
        __syncthreads();
        if (i < m) sums[i] = OP::f(carry, excl);
        carry = OP::f(carry, total);
    }
}
// out = scan of in with the tile's base from sums (EXCL: exclusive, else inclusive); in == out allowed
template <typename OP, bool EXCL>
__global__ __launch_bounds__(KNZ_RS_THREADS) void knz_scan_apply_kernel(const uint32_t* in, uint32_t* out, uint64_t n, const uint32_t* sums) {
    __shared__ uint32_t s_w[KNZ_RS_THREADS / 64];
    __shared__ uint32_t s_prev[KNZ_RS_THREADS];
    const uint64_t p0 = (uint64_t)blockIdx.x * KNZ_RS_TILE + (uint64_t)threadIdx.x * KNZ_RS_ITEMS;
    uint32_t v[KNZ_RS_ITEMS];
    uint32_t acc = 0;
    for (int j = 0; j < KNZ_RS_ITEMS; j++) { v[j] = p0 + j < n ? in[p0 + j] : 0u; acc = OP::f(acc, v[j]); }
    uint32_t total;
    const uint32_t incl = knz_wg256_scan_incl<OP>(acc, s_w, total);
    s_prev[threadIdx.x] = incl;
    __syncthreads();
    uint32_t run = OP::f(sums[blockIdx.x], threadIdx.x ? s_prev[threadIdx.x - 1] : 0u);
    for (int j = 0; j < KNZ_RS_ITEMS; j++) {
        if (p0 + j >= n) break;
        const uint32_t next = OP::f(run, v[j]);
        out[p0 + j] = EXCL ? run : next;
        run = next;
    }
}

// ---- select: out_idx[k] = k-th i with flags[i] != 0 (after the counts of the tiles went through the sum scan above) -------------------------
// bit j = flags[p0 + j] != 0 for the 16 flags of a thread (one 16-byte load when they are all inside the array and aligned)
__device__ __forceinline__ uint32_t knz_flags16(const uint8_t* flags, uint64_t p0, uint64_t n) {
    uint32_t f = 0;
    if (p0 + KNZ_RS_ITEMS <= n && (((uintptr_t)(flags + p0)) & 15) == 0) {
        const uint4 v = *(const uint4*)(flags + p0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int j = 0; j < 4; j++) f |= ((w[q] >> (8 * j)) & 0xFFu) ? 1u << (4 * q + j) : 0u;
    } else {
        for (int j = 0; j < KNZ_RS_ITEMS; j++) if (p0 + j < n && flags[p0 + j]) f |= 1u << j;
    }
    return f;
}
__global__ __launch_bounds__(KNZ_RS_THREADS) void knz_select_count_kernel(const uint8_t* flags, uint64_t n, uint32_t* sums) {
    __shared__ uint32_t s_w[KNZ_RS_THREADS / 64];
    const uint64_t p0 = (uint64_t)blockIdx.x * KNZ_RS_TILE + (uint64_t)threadIdx.x * KNZ_RS_ITEMS;
    const uint32_t f = knz_flags16(flags, p0, n);
    uint32_t total;
    knz_wg256_scan_incl<KnzOpSum>((uint32_t)__popc(f), s_w, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(KNZ_RS_THREADS) void knz_select_write_kernel(const uint8_t* flags, uint64_t n, const uint32_t* sums, uint32_t tilesN, uint32_t* out_idx, uint32_t* d_count) {
    __shared__ uint32_t s_w[KNZ_RS_THREADS / 64];
    __shared__ uint32_t s_prev[KNZ_RS_THREADS];
    const uint64_t p0 = (uint64_t)blockIdx.x * KNZ_RS_TILE + (uint64_t)threadIdx.x * KNZ_RS_ITEMS;
    const uint32_t f = knz_flags16(flags, p0, n), acc = (uint32_t)__popc(f);
    uint32_t total;
    const uint32_t incl = knz_wg256_scan_incl<KnzOpSum>(acc, s_w, total);
    s_prev[threadIdx.x] = incl;
    __syncthreads();
    uint32_t pos = sums[blockIdx.x] + (threadIdx.x ? s_prev[threadIdx.x - 1] : 0u);
    for (int j = 0; j < KNZ_RS_ITEMS; j++) if (f & (1u << j)) out_idx[pos++] = (uint32_t)(p0 + j);
    if (blockIdx.x == tilesN - 1 && threadIdx.x == KNZ_RS_THREADS - 1) *d_count = sums[blockIdx.x] + incl;
}

// ---- host drivers ---------------------------------------------------------------------------------------------------------------------------
// in-place exclusive sum scan of a flat u32 array of m elements; needs ceil(m / 4096) words of scratch
static int knz_own_scan_excl_sum(uint32_t* data, uint64_t m, uint32_t* sums, hipStream_t st) {
    const uint32_t t = (uint32_t)((m + KNZ_RS_TILE - 1) / KNZ_RS_TILE);
    hipLaunchKernelGGL(knz_scan_reduce_kernel<KnzOpSum>, dim3(t), dim3(KNZ_RS_THREADS), 0, st, (const uint32_t*)data, m, sums);
    hipLaunchKernelGGL(knz_scan_sums_kernel<KnzOpSum>, dim3(1), dim3(KNZ_RS_THREADS), 0, st, sums, t);
    hipLaunchKernelGGL((knz_scan_apply_kernel<KnzOpSum, true>), dim3(t), dim3(KNZ_RS_THREADS), 0, st, (const uint32_t*)data, data, m, (const uint32_t*)sums);
    return 0;
}

template <typename K>
static int knz_own_sort_pairs(DevBuf& tmp, K* kin, K* kout, uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1, hipStream_t st) {
    if (n == 0) return 0;
    if (n > 0xFFFFFFF0ull) return -1;
    if (b1 > 8 * sizeof(K) || b0 > b1) return -1;      // a digit beyond the key would shift by >= its width
    const unsigned passes = b1 > b0 ? (b1 - b0 + 7) / 8 : 0;
    if (passes == 0) {
        if (hipMemcpyAsync(kout, kin, n * sizeof(K), hipMemcpyDeviceToDevice, st) != hipSuccess || hipMemcpyAsync(vout, vin, n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return -1;
        return 0;
    }
    const uint32_t tiles = (uint32_t)((n + KNZ_RS_TILE - 1) / KNZ_RS_TILE);
    const uint64_t m = (uint64_t)256 * tiles;
    const size_t histBytes = ((size_t)m * 4 + 255) & ~(size_t)255, sumsBytes = ((size_t)((m + KNZ_RS_TILE - 1) / KNZ_RS_TILE) * 4 + 255) & ~(size_t)255;
    const bool third = (passes & 1) == 0;
    const size_t ktBytes = third ? ((n * sizeof(K) + 255) & ~(size_t)255) : 0, vtBytes = third ? ((n * 4 + 255) & ~(size_t)255) : 0;
    if (tmp.reserve(histBytes + sumsBytes + ktBytes + vtBytes + 256)) return -1;
    uint32_t* hist = tmp.as<uint32_t>();
    uint32_t* sums = (uint32_t*)(tmp.as<uint8_t>() + histBytes);
    K* kt = (K*)(tmp.as<uint8_t>() + histBytes + sumsBytes);
    uint32_t* vt = (uint32_t*)(tmp.as<uint8_t>() + histBytes + sumsBytes + ktBytes);
    const K* ks = kin; const uint32_t* vs = vin;
    for (unsigned p = 0; p < passes; p++) {
        const unsigned shift = b0 + 8 * p, nb = std::min(8u, b1 - shift);
        const uint32_t mask = (1u << nb) - 1;
        // destinations: out, in, out, in, ... ; with an even pass count the last but one goes to the third pair so that the last lands in out
        K* kd; uint32_t* vd;
        if (third && p == passes - 2) { kd = kt; vd = vt; }
        else if (p == passes - 1) { kd = kout; vd = vout; }
        else if ((p & 1) == 0) { kd = kout; vd = vout; }
        else { kd = kin; vd = vin; }
        hipLaunchKernelGGL(knz_rs_hist_kernel<K>, dim3(tiles), dim3(KNZ_RS_THREADS), 0, st, ks, (uint32_t)n, shift, mask, hist, tiles);
        knz_own_scan_excl_sum(hist, m, sums, st);
        hipLaunchKernelGGL(knz_rs_scatter_kernel<K>, dim3(tiles), dim3(KNZ_RS_THREADS), 0, st, ks, vs, kd, vd, (uint32_t)n, shift, mask, (const uint32_t*)hist, tiles);
        ks = kd; vs = vd;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

static int knz_own_scan_max_u32(DevBuf& tmp, uint32_t* in, uint32_t* out, size_t n, hipStream_t st) {
    if (n == 0) return 0;
    const uint32_t t = (uint32_t)((n + KNZ_RS_TILE - 1) / KNZ_RS_TILE);
    if (tmp.reserve((size_t)t * 4 + 256)) return -1;
    uint32_t* sums = tmp.as<uint32_t>();
    hipLaunchKernelGGL(knz_scan_reduce_kernel<KnzOpMax>, dim3(t), dim3(KNZ_RS_THREADS), 0, st, (const uint32_t*)in, (uint64_t)n, sums);
    hipLaunchKernelGGL(knz_scan_sums_kernel<KnzOpMax>, dim3(1), dim3(KNZ_RS_THREADS), 0, st, sums, t);
    hipLaunchKernelGGL((knz_scan_apply_kernel<KnzOpMax, false>), dim3(t), dim3(KNZ_RS_THREADS), 0, st, (const uint32_t*)in, out, (uint64_t)n, (const uint32_t*)sums);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

static int knz_own_select_flagged(DevBuf& tmp, const uint8_t* flags, uint32_t* out_idx, uint32_t* d_count, size_t n, hipStream_t st) {
    if (n == 0) return 0;
    const uint32_t t = (uint32_t)((n + KNZ_RS_TILE - 1) / KNZ_RS_TILE);
    if (tmp.reserve((size_t)t * 4 + 256)) return -1;
    uint32_t* sums = tmp.as<uint32_t>();
    hipLaunchKernelGGL(knz_select_count_kernel, dim3(t), dim3(KNZ_RS_THREADS), 0, st, flags, (uint64_t)n, sums);
    hipLaunchKernelGGL(knz_scan_sums_kernel<KnzOpSum>, dim3(1), dim3(KNZ_RS_THREADS), 0, st, sums, t);
    hipLaunchKernelGGL(knz_select_write_kernel, dim3(t), dim3(KNZ_RS_THREADS), 0, st, flags, (uint64_t)n, (const uint32_t*)sums, t, out_idx, d_count);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
