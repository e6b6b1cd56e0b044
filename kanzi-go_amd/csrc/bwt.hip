// Burrows-Wheeler transform on gfx950, batched over all blocks of a stream.
// Forward replaces BWTBlockCodec.Forward / BWT.Forward / DivSufSort.ComputeBWT (v2/transform/BWTBlockCodec.go:78-137,
// v2/transform/BWT.go:132-175, v2/transform/DivSufSort.go:179-311). The BWT is a function of the input alone (sorted
// suffixes, a suffix that ends first is smaller), so DivSufSort's induced sorting is replaced by a GPU suffix sort:
// prefix doubling over the concatenation of ALL blocks in HBM (block id in the top key bits keeps blocks apart):
//   round 0: radix sort of (block id, first 6 symbols) keys            [rocPRIM device radix sort, prims.h]
//   round k: only suffixes whose group is not yet a singleton are re-sorted by (group start, rank of suffix i+h);
//            group starts double as ranks, h doubles every round (Manber-Myers / Larsson-Sadakane refinement).
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

// round 0 keys: (block:10 bits | 6 symbols x 9 bits: symbol+1, 0 = past the end of the block)
__global__ __launch_bounds__(256) void knz_bwt_init_kernel(BwtGeom g, uint32_t total, uint64_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint32_t b = knz_bwt_block_of(g.gstart, g.nblocks, i);
    const uint32_t loc = i - g.gstart[b];
    const uint32_t n = g.in_len[b];
    const uint8_t* src = (const uint8_t*)g.in_ptr[b];
    uint64_t k = (uint64_t)b;
#pragma unroll
    for (int j = 0; j < 6; j++) k = (k << 9) | (loc + j < n ? (uint64_t)src[loc + j] + 1 : 0);
    keys[i] = k;
    vals[i] = i;
}

// group boundaries after a sort: head[j] = j if keys differ from the predecessor else 0 (max-scanned later)
__global__ __launch_bounds__(256) void knz_bwt_heads_kernel(const uint64_t* keys, uint32_t total, uint32_t* head) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    head[j] = (j == 0 || keys[j] != keys[j - 1]) ? j : 0;
}

// rank[SA[j]] = group start + 1 ; unresolved[j] = group has more than one member
__global__ __launch_bounds__(256) void knz_bwt_ranks_kernel(const uint32_t* sa, const uint32_t* gs, uint32_t total, uint32_t* rank, uint8_t* unresolved) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= total) return;
    const uint32_t s = gs[j];
    rank[sa[j]] = s + 1;
    const bool single = (s == j) && (j + 1 == total || gs[j + 1] == j + 1);
    unresolved[j] = single ? 0 : 1;
}

// subset keys for a doubling round: (group start, rank of suffix i+h inside the same block, 0 past the end)
__global__ __launch_bounds__(256) void knz_bwt_subkeys_kernel(BwtGeom g, const uint32_t* pos, uint32_t m, const uint32_t* sa, const uint32_t* gs,
                                                              const uint32_t* rank, uint32_t h, uint64_t* keys, uint32_t* vals) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    const uint32_t j = pos[k];
    const uint32_t i = sa[j];
    const uint32_t b = knz_bwt_block_of(g.gstart, g.nblocks, i);
    const uint32_t end = g.gstart[b + 1];
    const uint64_t r2 = (uint64_t)i + h < end ? rank[i + h] : 0;
    keys[k] = ((uint64_t)gs[j] << 32) | r2;
    vals[k] = i;
}

__global__ __launch_bounds__(256) void knz_bwt_subheads_kernel(const uint64_t* keys, const uint32_t* pos, uint32_t m, uint32_t* head) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    head[k] = (k == 0 || keys[k] != keys[k - 1]) ? pos[k] : 0;
}

// write the refined order back, new group starts / ranks / unresolved flags for the subset
__global__ __launch_bounds__(256) void knz_bwt_subupdate_kernel(const uint32_t* pos, uint32_t m, const uint32_t* vals, const uint32_t* gs2,
                                                                uint32_t* sa, uint32_t* gs, uint32_t* rank, uint8_t* unresolved) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    const uint32_t j = pos[k];
    const uint32_t s = gs2[k];
    sa[j] = vals[k];
    gs[j] = s;
    rank[vals[k]] = s + 1;
    const bool single = (s == j) && (k + 1 == m || gs2[k + 1] != s);
    unresolved[j] = single ? 0 : 1;
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
    const uint32_t p0 = a.rank[gs0] - 1 - gs0;        // local 0-based rank of suffix 0
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
            const uint32_t pi = a.rank[gs0 + c * step] - 1 - gs0;     // primaryIndex(c) - 1
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
    o.x = 0xFFFFFFFFu; o.y = e0.y;
    if (e0.x < cnt) {
        const uint2 e1 = in[gs0 + e0.x];
        o.x = e1.x < cnt ? e1.x : 0xFFFFFFFFu;
        o.y = e0.y | (e1.y << symBits);
    }
    out[p] = o;
}

// one wave per block, lanes 0..7 follow the 8 chunk chains (one lane when n < 256): 4 symbols per hop through links4,
// the last < 4 symbols of a chain through the single links
__global__ __launch_bounds__(64) void knz_bwt_inv_chains_kernel(BwtInvArgs a, const uint2* links, const uint2* links4) {
    const uint32_t b = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t* h = a.hdr + (size_t)b * 12;
    if (!a.active[b] || h[10] == 0) return;
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
