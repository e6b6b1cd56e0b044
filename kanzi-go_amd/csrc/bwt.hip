// Burrows-Wheeler transform on gfx950, batched over all blocks of a stream.
// Forward replaces BWTBlockCodec.Forward / BWT.Forward / DivSufSort.ComputeBWT (v2/transform/BWTBlockCodec.go:78-137,
// v2/transform/BWT.go:132-175, v2/transform/DivSufSort.go:179-311). The BWT is a function of the input alone (sorted
// suffixes, a suffix that ends first is smaller), so DivSufSort's induced sorting is replaced by a GPU suffix sort:
// prefix doubling over ALL blocks of the batch at once (bwt_sort.hip: per-block radix sort of the first symbols, then rounds that
// refine only the groups that are not singletons yet, by a segmented sort in LDS; Manber-Myers / Larsson-Sadakane refinement).
// Output rule (DivSufSort.go:187-197): dst[0] = src[n-1], then src[SA[r]-1] for every rank r except the rank of
// suffix 0; primaryIndex(k) = rank(suffix k*ceil(n/8)) + 1 (:202-206,:227-229,:283-285,:298-300,:309).
// Inverse replaces BWTBlockCodec.Inverse / BWT.inverseMergeTPSI / inverseBiPSIv2 (BWTBlockCodec.go:141-225,
// BWT.go:211-358,361-628): LF links by a stable sort on the symbol, then the (1|8) chains of a block run on 8 lanes.
#include "bits.h"

struct BwtGeom {
    uint32_t nblocks;
    const uint32_t* gstart;      // [nblocks+1] first global suffix index of each block (exclusive scan of lengths)
    const uint64_t* in_ptr;      // [nblocks]
    const uint32_t* in_len;      // [nblocks]
    const uint8_t* active;       // [nblocks] 1: block takes part
};

__device__ __forceinline__ uint32_t knz_bwt_block_of(const uint32_t* gstart, uint32_t nblocks, uint32_t g) {
    uint32_t lo = 0, hi = nblocks;            // largest b with gstart[b] <= g
    while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (gstart[mid] <= g) lo = mid; else hi = mid; }
    return lo;
}

struct BwtOutArgs {
    BwtGeom g;
    const uint32_t* sa; const uint32_t* rank;
    const uint64_t* out_ptr; uint32_t out_cap;
    uint32_t* out_len; int32_t* ok;
};

__device__ __forceinline__ void knz_bwt_header_geom(uint32_t n, uint32_t& pIndexSize, uint32_t& chunks, uint32_t& headerSize) {
    uint32_t lg = 31u - (uint32_t)__builtin_clz(n);
    if (n & (n - 1)) lg++;
    pIndexSize = (lg + 7) >> 3;                       // BWTBlockCodec.go:91-97
    chunks = n < 256 ? 1 : 8;                         // GetBWTChunks, BWT.go:631-637
    headerSize = chunks * pIndexSize + 1;
}

// BWT bytes (+ header written by thread 0 of each block's first workgroup)
__global__ __launch_bounds__(256) void knz_bwt_output_kernel(BwtOutArgs a, uint32_t total) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    const uint32_t b = knz_bwt_block_of(a.g.gstart, a.g.nblocks, j);
    const uint32_t gs0 = a.g.gstart[b];
    const uint32_t n = a.g.in_len[b];
    const uint8_t* src = (const uint8_t*)a.g.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    uint32_t pIndexSize, chunks, headerSize;
    knz_bwt_header_geom(n, pIndexSize, chunks, headerSize);
    const uint32_t r = j - gs0;
    const uint32_t p0 = a.rank[gs0] - 1;              // 0-based rank of suffix 0 (ranks are block-local, bwt_sort.hip)
    const uint32_t s = a.sa[j] - gs0;
    if (s != 0) dst[headerSize + (r < p0 ? r + 1 : r)] = src[s - 1];
    if (r == 0) {
        dst[headerSize] = src[n - 1];
        const uint32_t logNbChunks = chunks == 8 ? 3 : 0;
        dst[0] = (uint8_t)((logNbChunks << 2) | (pIndexSize - 1));
        uint32_t step = n / chunks;
        if (step * chunks != n) step++;
        uint32_t idx = 1;
        for (uint32_t c = 0; c < chunks; c++) {
            const uint32_t pi = a.rank[gs0 + c * step] - 1;           // primaryIndex(c) - 1
            for (int sh = (int)(pIndexSize - 1) * 8; sh >= 0; sh -= 8) dst[idx++] = (uint8_t)(pi >> sh);
        }
        a.out_len[b] = n + headerSize;
        a.ok[b] = 1;
    }
}

// blocks that cannot be transformed (BWTBlockCodec.Forward: n == 1 gives pIndexSize 0 -> "invalid index size")
__global__ void knz_bwt_precheck_kernel(uint32_t nblocks, const uint32_t* in_len, const uint8_t* active, uint32_t out_cap, uint8_t* take, int32_t* ok) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t n = in_len[b];
    const bool t = active[b] && n >= 2 && out_cap >= n + 33;
    take[b] = t ? 1 : 0;
    ok[b] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Inverse
struct BwtInvArgs {
    uint32_t nblocks;
    const uint32_t* gstart;        // [nblocks+1] over the BWT payload bytes (header excluded)
    const uint64_t* in_ptr;        // [nblocks] block data (header first)
    const uint32_t* in_len;        // [nblocks] incl. header
    const uint8_t* active;
    uint32_t* hdr;                 // [nblocks*12] {headerSize, chunks, pidx[8], valid}
    const uint64_t* out_ptr; uint32_t out_cap; uint32_t* out_len; int32_t* ok;
};

// header parse + validity (BWTBlockCodec.go:156-190) ; payload length -> plen[b]
__global__ void knz_bwt_inv_header_kernel(BwtInvArgs a, uint32_t* plen) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks) return;
    uint32_t* h = a.hdr + (size_t)b * 12;
    plen[b] = 0;
    h[10] = 0;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    int32_t err = 0;
    if (n == 1) err = KNZ_ERR_PROCESS_BLOCK;
    uint32_t headerSize = 0, chunks = 0;
    if (!err) {
        const uint32_t mode = src[0];
        const uint32_t logNbChunks = (mode >> 2) & 7, pIndexSize = (mode & 3) + 1;
        chunks = 1u << logNbChunks;
        headerSize = chunks * pIndexSize + 1;
        if (n < headerSize) err = KNZ_ERR_PROCESS_BLOCK;
        else if (chunks != ((n - headerSize) < 256 ? 1u : 8u)) err = KNZ_ERR_PROCESS_BLOCK;
        else {
            uint32_t idx = 1;
            for (uint32_t c = 0; c < chunks; c++) {
                uint32_t pi = 0;
                for (uint32_t k = 0; k < pIndexSize; k++) pi = (pi << 8) | src[idx++];
                h[2 + c] = pi + 1;                           // primaryIndex
            }
            const uint32_t cnt = n - headerSize;
            if (cnt > a.out_cap) err = KNZ_ERR_PROCESS_BLOCK;
            if (cnt >= 2) {
                if (h[2] == 0 || h[2] > cnt) err = KNZ_ERR_PROCESS_BLOCK;     // corrupted primary index (BWT.go:216-220)
                for (uint32_t c = 0; c < chunks; c++) if (h[2 + c] == 0 || h[2 + c] > cnt) err = KNZ_ERR_PROCESS_BLOCK;
            }
        }
    }
    h[0] = headerSize; h[1] = chunks;
    if (err) { a.ok[b] = -err; a.out_len[b] = 0; return; }
    h[10] = 1;
    plen[b] = n - headerSize;
    a.out_len[b] = n - headerSize;
    a.ok[b] = 1;
}

// sort keys: (block << 8 | symbol), value = global payload index
__global__ __launch_bounds__(256) void knz_bwt_inv_keys_kernel(BwtInvArgs a, uint32_t total, uint32_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint32_t b = knz_bwt_block_of(a.gstart, a.nblocks, i);
    const uint32_t loc = i - a.gstart[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b] + a.hdr[(size_t)b * 12];
    keys[i] = (b << 8) | src[loc];
    vals[i] = i;
}

// Distance between the splitters of the inverse's list ranking (below). A PRIME: with a splitter on every 128th slot the walks of record-shaped and
// executable-like blocks resonated with their power-of-two periods (a walk that advances by multiples of 128 slots meets no splitter: the longest
// sub-lists set the time of the walk and emit kernels); round 6, S-silesia 26 x 8 MiB: walk 19.7 -> 6.8 ms, emit 6.2 -> 1.9 ms, ranking 2.8 -> 1.7 ms.
#ifndef KNZ_BWT_SPLIT
#define KNZ_BWT_SPLIT 191u
#endif

// LF links (BWT.go:228-247): slot p of the stably sorted order holds symbol v from payload index i; the link is
// i-1 for 1 <= i < pIdx, i for i >= pIdx (the entry of i = 0 ends the text and is never followed).
__global__ __launch_bounds__(256) void knz_bwt_inv_links_kernel(BwtInvArgs a, uint32_t total, const uint32_t* skeys, const uint32_t* svals, uint2* links) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const uint32_t b = skeys[p] >> 8;
    const uint32_t gs0 = a.gstart[b];
    const uint32_t i = svals[p] - gs0;
    const uint32_t pIdx = a.hdr[(size_t)b * 12 + 2];
    uint2 e;
    e.x = i == 0 ? 0xFFFFFFFFu : (i < pIdx ? i - 1 : i);
    e.y = skeys[p] & 0xFF;
    // bit 31: the slot this link leads to is a splitter of the list ranking below (a multiple of KNZ_BWT_SPLIT or the start of
    // one of the block's chunk chains): a walker learns it has arrived without another memory access
    if (e.x != 0xFFFFFFFFu) {
        bool sp = (e.x % KNZ_BWT_SPLIT) == 0;
        const uint32_t* h = a.hdr + (size_t)b * 12;
        for (uint32_t c = 0; c < h[1]; c++) sp = sp || (h[2 + c] - 1 == e.x);
        if (sp) e.y |= 0x80000000u;
    }
    links[p] = e;
}

// Pointer doubling of the LF links: out[p] = two hops of in[p] with the symbols of both packed little-endian. Applied twice
// (1 -> 2 -> 4 symbols per hop, an entry stays 8 bytes) it cuts the dependent pointer chase of every chain by 4 for two fully
// parallel gather passes. A hop that leaves the block (corrupt stream, or the end of the permutation cycle) makes the composite
// link invalid (0xFFFFFFFF); the chains only follow composites while at least that many symbols are still due.
__global__ __launch_bounds__(256) void knz_bwt_inv_double_kernel(BwtInvArgs a, uint32_t total, const uint32_t* skeys, const uint2* in, uint2* out, uint32_t symBits) {
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const uint32_t b = skeys[p] >> 8;
    const uint32_t gs0 = a.gstart[b];
    const uint32_t cnt = a.in_len[b] - a.hdr[(size_t)b * 12];
    const uint2 e0 = in[p];
    uint2 o;
    const uint32_t ymask = symBits == 8 ? 0xFFu : 0xFFFFFFFFu;    // (the single links carry the splitter flag in bit 31)
    o.x = 0xFFFFFFFFu; o.y = e0.y & ymask;
    if (e0.x < cnt) {
        const uint2 e1 = in[gs0 + e0.x];
        o.x = e1.x < cnt ? e1.x : 0xFFFFFFFFu;
        o.y = (e0.y & ymask) | ((e1.y & ymask) << symBits);
    }
    out[p] = o;
}

// one wave per block, lanes 0..7 follow the 8 chunk chains (one lane when n < 256): 4 symbols per hop through links4,
// the last < 4 symbols of a chain through the single links
__global__ __launch_bounds__(64) void knz_bwt_inv_chains_kernel(BwtInvArgs a, const uint2* links, const uint2* links4, const uint8_t* use_chain) {
    const uint32_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t* h = a.hdr + (size_t)b * 12;
    if (!a.active[b] || h[10] == 0) return;
    if (use_chain && !use_chain[b]) return;                          // the block went through the list ranking
    const uint32_t cnt = a.in_len[b] - h[0];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b] + h[0];
    if (cnt == 1) { if (lane == 0) dst[0] = src[0]; return; }
    const uint2* L = links + a.gstart[b];
    const uint2* L4 = links4 + a.gstart[b];
    const uint32_t chunks = h[1];
    if ((uint32_t)lane >= chunks) return;
    uint32_t ck = chunks == 8 ? (cnt >> 3) : cnt;
    if (chunks == 8 && ck * 8 != cnt) ck++;
    const uint32_t start = (uint32_t)lane * ck;
    const uint32_t end = min(cnt, start + ck);
    uint32_t t = h[2 + lane] - 1;
    uint32_t i = start;
    for (; i + 4 <= end; i += 4) {
        if (t >= cnt) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; return; }
        const uint2 e = L4[t];
        dst[i] = (uint8_t)e.y; dst[i + 1] = (uint8_t)(e.y >> 8); dst[i + 2] = (uint8_t)(e.y >> 16); dst[i + 3] = (uint8_t)(e.y >> 24);
        if (e.x == 0xFFFFFFFFu) {
            // the composite left the block. Legitimate only as the link behind the chain's very last symbol (the last chain
            // ends on the terminator): the 3 hops inside must still be good.
            uint32_t q = t;
            for (int hop = 0; hop < 3; hop++) { q = L[q].x; if (q >= cnt) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; return; } }
        }
        t = e.x;
    }
    for (; i < end; i++) {                                           // < 4 symbols left: single links
        if (t >= cnt) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; return; }
        const uint2 e = L[t];
        dst[i] = (uint8_t)e.y;
        t = e.x;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Inverse BWT without the block-long pointer chase: LIST RANKING with splitters (Helman-JaJa). The LF links make one linked
// list of n slots per block; the 8 chunk chains the format allows are still 1 M dependent HBM accesses each at 8 MiB (~430 ns per
// hop, profiles/r02_lone_wave_latencies.md: 160 ms whatever the GPU does meanwhile). Instead:
//   1. every KNZ_BWT_SPLIT-th slot (and every chunk start) is a splitter; one THREAD per splitter walks the list to the next
//      splitter and records (successor, length): ~191 hops each, n / 191 independent walks per block: throughput, not latency;
//   2. one workgroup per block ranks its splitter list by pointer jumping (Wyllie, log2(n/191) rounds over 44 K entries in L2):
//      distance of every splitter to the end of the text, hence its text position;
//   3. one thread per splitter walks its sub-list again, four symbols per hop through the doubled links, and stores the symbols
//      at the text positions now known.
// A block whose list is not one path from the first primary index to the terminator through all n slots with the chunk starts
// at their positions (damaged stream), or with a sub-list longer than the cap, is handed to the chains kernel, which keeps the
// reference's behaviour on such input. Blocks below KNZ_BWT_RANK_MIN go there directly.
#define KNZ_BWT_SP_END 0xFFFFFFFFu
#define KNZ_BWT_SP_UNUSED 0xFFFFFFFEu
#define KNZ_BWT_WALK_CAP (1u << 22)

struct BwtRankArgs {
    BwtInvArgs a;
    const uint2* links; const uint2* links4;
    const uint32_t* sp_base;       // [nblocks + 1] first splitter id of each block (0 splitters: block takes the chains kernel)
    uint32_t* sp_succ; uint32_t* sp_len;      // [total splitters]
    uint32_t* sp_succ2; uint32_t* sp_dist; uint32_t* sp_dist2;
    uint32_t* sp_pos;              // text position of the splitter's first symbol
    uint8_t* use_chain;            // [nblocks] 1 = the chains kernel decodes this block
};

__device__ __forceinline__ uint32_t knz_bwt_sp_id(const uint32_t* h, uint32_t nreg, uint32_t slot) {
    if (slot % KNZ_BWT_SPLIT == 0) return slot / KNZ_BWT_SPLIT;
    for (uint32_t c = 0; c < h[1]; c++) if (h[2 + c] - 1 == slot) return nreg + c;
    return KNZ_BWT_SP_UNUSED;
}

__global__ __launch_bounds__(256) void knz_bwt_inv_walk_kernel(BwtRankArgs r, uint32_t total_sp) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= total_sp) return;
    const uint32_t b = knz_bwt_block_of(r.sp_base, r.a.nblocks, k);
    const uint32_t j = k - r.sp_base[b];
    const uint32_t* h = r.a.hdr + (size_t)b * 12;
    const uint32_t cnt = r.a.in_len[b] - h[0];
    const uint32_t nreg = (cnt + KNZ_BWT_SPLIT - 1) / KNZ_BWT_SPLIT;
    uint32_t slot;
    if (j < nreg) slot = j * KNZ_BWT_SPLIT;
    else {
        const uint32_t c = j - nreg;
        slot = c < h[1] ? h[2 + c] - 1 : 0xFFFFFFFFu;
        if (slot >= cnt || slot % KNZ_BWT_SPLIT == 0) { r.sp_succ[k] = KNZ_BWT_SP_UNUSED; r.sp_len[k] = 0; return; }   // (a chunk start that is a regular splitter already)
    }
    const uint2* L = r.links + r.a.gstart[b];
    uint32_t t = slot, len = 0, succ = KNZ_BWT_SP_END;
    for (;;) {
        const uint2 e = L[t];
        len++;
        if (e.x >= cnt) break;                                       // the terminator's link (or a damaged one): end of this sub-list
        if (e.y & 0x80000000u) { succ = knz_bwt_sp_id(h, nreg, e.x); break; }
        t = e.x;
        if (len >= KNZ_BWT_WALK_CAP) { succ = KNZ_BWT_SP_UNUSED; break; }   // no splitter in reach: the rank kernel rejects the block
    }
    r.sp_succ[k] = succ;
    r.sp_len[k] = len;
}

__global__ __launch_bounds__(1024) void knz_bwt_inv_rank_kernel(BwtRankArgs r) {
    __shared__ int s_bad;
    const uint32_t b = blockIdx.x;
    const uint32_t base = r.sp_base[b], n = r.sp_base[b + 1] - base;
    if (n == 0) return;
    const uint32_t* h = r.a.hdr + (size_t)b * 12;
    if (!r.a.active[b] || h[10] == 0) return;
    const uint32_t cnt = r.a.in_len[b] - h[0];
    const uint32_t nreg = (cnt + KNZ_BWT_SPLIT - 1) / KNZ_BWT_SPLIT;
    const uint32_t tid = threadIdx.x;
    uint32_t* sa = r.sp_succ + base; uint32_t* sb = r.sp_succ2 + base;
    uint32_t* da = r.sp_dist + base; uint32_t* db = r.sp_dist2 + base;
    if (tid == 0) s_bad = 0;
    __syncthreads();                                                          // (the flag is cleared before any thread can raise it)
    for (uint32_t i = tid; i < n; i += 1024) {
        const uint32_t s = sa[i];
        da[i] = r.sp_len[base + i];
        if (s == KNZ_BWT_SP_UNUSED && r.sp_len[base + i] != 0) s_bad = 1;        // a walk that found no splitter / an unknown successor
    }
    __syncthreads();
    uint32_t rounds = 1;
    while ((1u << rounds) < n) rounds++;
    for (uint32_t it = 0; it <= rounds; it++) {
        for (uint32_t i = tid; i < n; i += 1024) {
            const uint32_t s = sa[i];
            if (s < n) { sb[i] = sa[s]; db[i] = da[i] + da[s]; }
            else { sb[i] = s; db[i] = da[i]; }
        }
        __syncthreads();
        uint32_t* t0 = sa; sa = sb; sb = t0;
        uint32_t* t1 = da; da = db; db = t1;
    }
    // every used splitter must have reached the end of the text (cycles never do), the list must hold all cnt slots and every
    // chunk must start where the format says
    for (uint32_t i = tid; i < n; i += 1024)
        if (r.sp_len[base + i] != 0 && sa[i] != KNZ_BWT_SP_END) s_bad = 1;
    __syncthreads();
    const uint32_t start = knz_bwt_sp_id(h, nreg, h[2] - 1);
    if (start >= n) { if (tid == 0) r.use_chain[b] = 1; return; }
    const uint32_t total = da[start];
    if (tid == 0) {
        if (total != cnt) s_bad = 1;
        uint32_t ck = h[1] == 8 ? (cnt >> 3) : cnt;
        if (h[1] == 8 && ck * 8 != cnt) ck++;
        for (uint32_t c = 0; c < h[1]; c++) {
            const uint32_t id = knz_bwt_sp_id(h, nreg, h[2 + c] - 1);
            if (id >= n || total - da[id] != c * ck) s_bad = 1;
        }
    }
    __syncthreads();
    if (s_bad) { if (tid == 0) r.use_chain[b] = 1; return; }
    for (uint32_t i = tid; i < n; i += 1024) r.sp_pos[base + i] = total - da[i];
}

__global__ __launch_bounds__(256) void knz_bwt_inv_emit_kernel(BwtRankArgs r, uint32_t total_sp) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= total_sp) return;
    const uint32_t b = knz_bwt_block_of(r.sp_base, r.a.nblocks, k);
    if (r.use_chain[b]) return;
    const uint32_t j = k - r.sp_base[b];
    uint32_t len = r.sp_len[k];
    if (len == 0) return;
    const uint32_t* h = r.a.hdr + (size_t)b * 12;
    if (!r.a.active[b] || h[10] == 0) return;
    const uint32_t cnt = r.a.in_len[b] - h[0];
    const uint32_t nreg = (cnt + KNZ_BWT_SPLIT - 1) / KNZ_BWT_SPLIT;
    uint32_t t = j < nreg ? j * KNZ_BWT_SPLIT : h[2 + (j - nreg)] - 1;
    const uint2* L = r.links + r.a.gstart[b];
    const uint2* L4 = r.links4 + r.a.gstart[b];
    uint8_t* dst = (uint8_t*)r.a.out_ptr[b] + r.sp_pos[k];
    for (; len >= 4; len -= 4, dst += 4) {                            // (validated: all len slots of the sub-list are inside the block)
        const uint2 e = L4[t];
        dst[0] = (uint8_t)e.y; dst[1] = (uint8_t)(e.y >> 8); dst[2] = (uint8_t)(e.y >> 16); dst[3] = (uint8_t)(e.y >> 24);
        t = e.x;
    }
    for (; len > 0; len--, dst++) {
        const uint2 e = L[t];
        dst[0] = (uint8_t)e.y;
        t = e.x;
    }
}
