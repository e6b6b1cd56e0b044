// Static Huffman encoder of kanzi bitstream v6 as one gfx950 kernel: one 256-thread workgroup (4 wave64)
// per 16 KiB chunk.  Replaces HuffmanEncoder.Write / updateFrequencies / computeCodeLengths /
// limitCodeLengths / generateCanonicalCodes / encodeChunk (v2/entropy/HuffmanCodec.go:390,128,300,216,37,435)
// and internal.ComputeHistogram order 0 (v2/internal/Global.go:226-251).
//
// Per chunk the kernel emits 5 bit-string units into a fixed-stride scratch slot (bits.h):
//   u0 = alphabet + Exp-Golomb code-length deltas + 4 varint fragment bit counts (or the raw bytes of a
//        <32-byte chunk, HuffmanCodec.go:411-413), u1..u3 = fragments 0..2, u4 = fragment 3 + (n&3) tail bytes.
// A later scan + gather (layout.hip) places the units at their final bit positions, which is the device
// form of the bit-granular WriteArray concatenation (v2/bitstream/DefaultOutputBitStream.go:101-199).
//
// LDS: 4 x 6160 B fragment staging (worst case 12 bits/symbol), 4 privatised histograms, code tables.
// HBM traffic per chunk: n bytes read twice (histogram pass coalesced 16 B/lane, encode pass row-per-lane;
// the second read is an L2 hit), compressed bytes written once.
#include "bits.h"

// ---- EntropyUtils.go:123-260 NormalizeFrequencies, used only by limitCodeLengths' slow path --------------
// freqs[0..n) and alphabet[0..n) in private memory; returns alphabet size. Go's index-out-of-range panic on
// the totalFreq == scale shortcut with n < 256 is reported through *panic.
__device__ static int knz_normalize_freqs(int* freqs, int n, int* alphabet, int totalFreq, int scale, int* panic) {
    if (n == 0 || totalFreq == 0) return 0;
    int alphabetSize = 0;
    if (totalFreq == scale) {
        if (n < 256) { *panic = 1; return 0; }
        for (int i = 0; i < 256; i++) if (freqs[i] != 0) alphabet[alphabetSize++] = i;
        return alphabetSize;
    }
    int sumScaledFreq = 0, sumFreq = 0, idxMax = 0;
    for (int i = 0; i < n; i++) {
        alphabet[i] = 0;
        int f = freqs[i];
        if (f == 0) continue;
        long long sf = (long long)f * scale;
        int scaledFreq;
        if (sf <= totalFreq) scaledFreq = 1;
        else scaledFreq = (int)((sf + (totalFreq >> 1)) / totalFreq);
        alphabet[alphabetSize++] = i;
        sumScaledFreq += scaledFreq;
        freqs[i] = scaledFreq;
        sumFreq += f;
        if (scaledFreq > freqs[idxMax]) idxMax = i;
        if (sumFreq >= totalFreq) break;
    }
    if (alphabetSize == 0) return 0;
    if (alphabetSize == 1) { freqs[alphabet[0]] = scale; return 1; }
    if (sumScaledFreq == scale) return alphabetSize;
    int delta = sumScaledFreq - scale;
    int errThr = freqs[idxMax] >> 4;
    int inc;
    int absDelta = delta < 0 ? -delta : delta;
    if (absDelta <= errThr) { freqs[idxMax] -= delta; return alphabetSize; }
    if (delta < 0) { delta += errThr; freqs[idxMax] += errThr; inc = 1; delta = -delta; }
    else { delta -= errThr; freqs[idxMax] -= errThr; inc = -1; }
    int round = 1;
    while (round < 6 && delta > 0) {
        int adjustments = 0;
        round++;
        for (int k = 0; k < alphabetSize; k++) {
            int idx = alphabet[k];
            if (freqs[idx] <= 2) continue;
            freqs[idx] += inc;
            adjustments++;
            delta--;
            if (delta == 0) break;
        }
        if (adjustments == 0) break;
    }
    int v = freqs[idxMax] - delta;
    freqs[idxMax] = v > 1 ? v : 1;
    return alphabetSize;
}

// ---- HuffmanCodec.go:328-385 Moffat-Katajainen in-place code lengths (serial, one thread) ----------------
__device__ static int knz_huf_lengths_inplace(uint32_t* data, int n) {
    // phase 1 (:328-356)
    for (int s = 0, r = 0, t = 0; t < n - 1; t++) {
        uint32_t sum = 0;
        for (int i = 0; i < 2; i++) {
            if (s >= n || (r < t && data[r] < data[s])) {
                sum += data[r];
                data[r] = (uint32_t)t;
                r++;
                continue;
            }
            sum += data[s];
            if (s > t) data[s] = 0;
            s++;
        }
        data[t] = sum;
    }
    // phase 2 (:359-385)
    int levelTop = n - 2, depth = 1, i = n, totalNodesAtLevel = 2;
    while (i > 0) {
        int k = levelTop;
        while (k > 0 && (int)data[k - 1] >= levelTop) k--;
        int internalNodesAtLevel = levelTop - k;
        int leavesAtLevel = totalNodesAtLevel - internalNodesAtLevel;
        for (int j = 0; j < leavesAtLevel; j++) { i--; data[i] = (uint32_t)depth; }
        totalNodesAtLevel = internalNodesAtLevel << 1;
        levelTop = k;
        depth++;
    }
    return depth - 1;
}

// Serial sort of keys (only used by the rare slow path; the main path sorts in parallel by rank counting)
__device__ static void knz_insertion_sort(uint32_t* a, int n) {
    for (int i = 1; i < n; i++) {
        uint32_t v = a[i];
        int j = i - 1;
        while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; j--; }
        a[j + 1] = v;
    }
}

// ---- HuffmanCodec.go:216-297 limitCodeLengths (serial, rare) ------------------------------------------------
// sorted[i] = symbol of rank i (increasing frequency), lens[sym] current lengths, freq[sym] chunk histogram.
// alpha[i] = i-th symbol of the alphabet in increasing symbol order. Returns max code length (>12 => fallback).
// tmp: >= 3584 bytes of LDS that nothing else uses yet (the fragment staging area), so the rare path costs
// no per-lane scratch memory.
__device__ static int knz_huf_limit(const uint8_t* alpha, int count, const uint32_t* freq, uint8_t* lens,
                                    uint32_t* sorted, uint32_t* data, int* panic, uint32_t* tmp) {
    int n = 0, debt = 0;
    while (lens[sorted[n]] >= KNZ_HUF_MAXLEN) {
        debt += (int)lens[sorted[n]] - KNZ_HUF_MAXLEN;
        lens[sorted[n]] = KNZ_HUF_MAXLEN;
        n++;
        if (n >= count) { *panic = 1; return KNZ_HUF_MAXLEN; }
    }
    uint8_t (*q)[256] = (uint8_t (*)[256])tmp;           // 1536 bytes
    int* f = (int*)(tmp + 384);                             // 1024 bytes
    int* al = (int*)(tmp + 640);                            // 1024 bytes
    int qh[6], qt[6];
    for (int i = 0; i < 6; i++) { qh[i] = 0; qt[i] = 0; }
    while (n < count) {
        int idx = KNZ_HUF_MAXLEN - 1 - (int)lens[sorted[n]];
        if (idx > 5 || debt < (1 << idx)) break;
        q[idx][qt[idx]++] = (uint8_t)sorted[n];
        n++;
    }
    int idx = 5;
    while (debt > 0 && idx >= 0) {
        if (qh[idx] == qt[idx] || debt < (1 << idx)) { idx--; continue; }
        lens[q[idx][qh[idx]++]]++;
        debt -= (1 << idx);
    }
    idx = 0;
    while (debt > 0 && idx < 6) {
        if (qh[idx] == qt[idx]) { idx++; continue; }
        lens[q[idx][qh[idx]++]]++;
        debt -= (1 << idx);
    }
    if (debt > 0) {
        // slow path (:273-294): renormalise to 2048 and recompute
        int totalFreq = 0;
        for (int i = 0; i < count; i++) { f[i] = (int)freq[alpha[i]]; totalFreq += f[i]; }
        knz_normalize_freqs(f, count, al, totalFreq, KNZ_HUF_CHUNK >> 3, panic);
        if (*panic) return KNZ_HUF_MAXLEN;
        for (int i = 0; i < count; i++) sorted[i] = ((uint32_t)f[i] << 8) | alpha[i];
        knz_insertion_sort(sorted, count);
        for (int i = 0; i < count; i++) {
            data[i] = sorted[i] >> 8;
            sorted[i] &= 0xFF;
            if (data[i] == 0) { *panic = 1; return KNZ_HUF_MAXLEN; }
        }
        int maxLen = knz_huf_lengths_inplace(data, count);
        for (int i = 0; i < count; i++) lens[sorted[i]] = (uint8_t)data[i];
        return maxLen;
    }
    return KNZ_HUF_MAXLEN;
}

// Parameters of one encode batch. Chunk (b,k) covers post-transform bytes [k*16384, ...) of block b.
struct HufEncArgs {
    const uint8_t* data;          // post-transform bytes of all blocks
    const uint64_t* blk_off;      // [nblocks] byte offset of block b inside data
    const uint32_t* blk_len;      // [nblocks] post-transform length of block b
    uint32_t chunks_per_block;    // CPB: chunk slots reserved per block
    uint8_t* scratch;             // [(nblocks*CPB) * KNZ_CHUNK_STRIDE]
    uint32_t* unit_bits;          // [(nblocks*CPB) * 5]
    uint32_t* unit_src;           // [(nblocks*CPB) * 5] byte offset of each unit inside its scratch slot
    int32_t* blk_status;          // [nblocks] set to ERR_PROCESS_BLOCK (13) where the Go code would panic
    // sorted statistics between the three kernels, transposed per group of 64 chunks: entry (chunk c, rank i) at
    // ((c / 64) * 256 + i) * 64 + c % 64, so that a wave working on 64 chunks (one per lane) reads and writes coalesced
    uint16_t* st_freq;            // frequency of the symbol of rank i (ascending (frequency, symbol) order)
    uint8_t* st_sym;              // that symbol
    uint8_t* st_len;              // its code length (Moffat-Katajainen, before the 12-bit limiter)
    uint16_t* st_count;           // [chunks] alphabet size (0: chunk absent or stored raw)
    uint8_t* st_maxlen;           // [chunks] longest code length
    uint32_t nchunks;
    // Encode at the final bit positions (round 4): the histogram kernel also leaves the symbol counts of each of the chunk's four fragments,
    // knz_huf_encode_kernel<true> turns them into the exact bit count of every unit WITHOUT encoding (sum of count x code length, the header built
    // as it will be), the layout kernels scan those, and knz_huf_encode_kernel<false> then writes its units straight to where the stream has them
    // (dst_words != null): the compressed bytes no longer make a round trip through the scratch slots and knz_gather_kernel.
    uint16_t* st_fhist;           // [chunks][4][256] symbol counts per fragment (null: not wanted)
    uint32_t* dst_words;          // the stream (null: units go to the scratch slots)
    const uint64_t* chunk_rel; const uint64_t* blk_dst_bit; const uint64_t* total_bits;
};

// 32 bits of an MSB-first bit string held as 32-bit words in LDS, from bit `off` of the string (negative: the string starts inside the word);
// bits at or behind `len` read as zero (the words behind the string are not touched)
__device__ __forceinline__ uint32_t knz_lds_fetch32(const uint32_t* w, int64_t off, int64_t len) {
    if (off <= -32 || off >= len || len <= 0) return 0;
    uint32_t lead = 0;
    if (off < 0) { lead = (uint32_t)(-off); off = 0; }
    const uint32_t i = (uint32_t)(off >> 5), sh = (uint32_t)off & 31;
    uint32_t v = w[i] << sh;
    if (sh && (int64_t)(i + 1) * 32 < len) v |= w[i + 1] >> (32 - sh);
    const int64_t avail = len - off;
    if (avail < 32) v &= 0xFFFFFFFFu << (32 - (uint32_t)avail);
    return v >> lead;
}

__device__ __forceinline__ size_t knz_huf_st(uint32_t chunk, uint32_t i) { return ((size_t)(chunk >> 6) * 256 + i) * 64 + (chunk & 63); }

// ---- kernel 1: histogram (Global.go:226-251) + sort of (frequency << 8 | symbol) (sort.Ints in computeCodeLengths :303) ----
// 256 threads per chunk, thread = symbol. The rank of a key is the number of smaller keys: every wave holds the 256 keys in
// 4 registers (key 64*w + lane in register w) and walks them with v_readlane, no LDS in the loop.
__global__ __launch_bounds__(256) void knz_huf_hist_kernel(HufEncArgs a) {
    __shared__ uint32_t s_hist[16][256];
    __shared__ uint32_t s_key[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t cpb = a.chunks_per_block;
    const uint32_t b = blockIdx.x / cpb, k = blockIdx.x % cpb;
    const uint32_t postLen = a.blk_len[b];
    uint32_t n = 0;
    if ((uint64_t)k * KNZ_HUF_CHUNK < postLen) n = min((uint32_t)KNZ_HUF_CHUNK, postLen - k * KNZ_HUF_CHUNK);
    if (n < 32) {                                             // absent or raw (HuffmanCodec.go:411-413)
        if (tid == 0) { a.st_count[blockIdx.x] = 0; a.st_maxlen[blockIdx.x] = 0; }
        return;
    }
    const uint8_t* src = a.data + a.blk_off[b] + (size_t)k * KNZ_HUF_CHUNK;
    for (int i = tid; i < 16 * 256; i += 256) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    // wave j counts fragment j = symbols [j * F, (j + 1) * F), F = n / 4 (encodeChunk :435-492), into its own four histograms: their sum is the
    // fragment's symbol count (the sizes pass needs it), the sum over the waves plus the n & 3 tail bytes the chunk's
    const uint32_t F = n >> 2;
    knz_histogram_64t_x4(src + (size_t)(tid >> 6) * F, F, s_hist + 4 * (tid >> 6), tid & 63);
    __syncthreads();
    uint32_t myFreq = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t fj = s_hist[4 * j][tid] + s_hist[4 * j + 1][tid] + s_hist[4 * j + 2][tid] + s_hist[4 * j + 3][tid];
        if (a.st_fhist) a.st_fhist[((size_t)blockIdx.x * 4 + j) * 256 + tid] = (uint16_t)fj;      // <= 4096
        myFreq += fj;
    }
    for (uint32_t i = 4 * F; i < n; i++) myFreq += src[i] == (uint32_t)tid ? 1u : 0u;
    const uint32_t myKey = myFreq ? ((myFreq << 8) | (uint32_t)tid) : 0xFFFFFFFFu;
    s_key[tid] = myKey;
    __syncthreads();
    uint32_t kreg[4];
#pragma unroll
    for (int w = 0; w < 4; w++) kreg[w] = s_key[64 * w + lane];
    uint32_t rank = 0, count = 0;
    for (uint32_t j = 0; j < 64; j++) {
#pragma unroll
        for (int w = 0; w < 4; w++) {
            const uint32_t kj = wave_readlane(kreg[w], j);
            rank += kj < myKey ? 1u : 0u;
            count += kj != 0xFFFFFFFFu ? 1u : 0u;
        }
    }
    if (myFreq) {
        a.st_freq[knz_huf_st(blockIdx.x, rank)] = (uint16_t)myFreq;     // <= 16384
        a.st_sym[knz_huf_st(blockIdx.x, rank)] = (uint8_t)tid;
    }
    if (tid == 0) a.st_count[blockIdx.x] = (uint16_t)count;
}

// ---- kernel 2: Moffat-Katajainen in-place code lengths (HuffmanCodec.go:328-385), ONE LANE PER CHUNK --------------------
// The algorithm is a serial chain of dependent reads; 64 chunks per wave keep the LDS pipe busy instead of one lane of a
// 256-thread workgroup. Array element i of lane l lives at s_d[i * 64 + l] (bank = lane: conflict free when the lanes move
// in step, which they mostly do).
__global__ __launch_bounds__(64) void knz_huf_lengths_kernel(HufEncArgs a) {
    __shared__ uint16_t s_d[256 * 64];
    const int lane = threadIdx.x;
    const uint32_t chunk = blockIdx.x * 64 + lane;
    const int n = chunk < a.nchunks ? (int)a.st_count[chunk] : 0;
    uint16_t* d = s_d + lane;
    {   // the sorted frequencies, 8 loads in flight per lane (one load per trip left the chain waiting on HBM 256 times)
        const int nmax = (int)wave_reduce_max((uint32_t)n);
        const uint32_t cc = chunk < a.nchunks ? chunk : 0u;
        for (int i0 = 0; i0 < nmax; i0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = i0 + u < n ? a.st_freq[knz_huf_st(cc, i0 + u)] : 0u;
#pragma unroll
            for (int u = 0; u < 8; u++) if (i0 + u < n) d[(i0 + u) * 64] = (uint16_t)v[u];
        }
    }
    if (n < 2) {                                              // 0: nothing, 1: the single symbol gets length 1 (:156-158)
        if (n == 1) { a.st_len[knz_huf_st(chunk, 0)] = 1; a.st_maxlen[chunk] = 1; }
        return;
    }
    // phase 1 (:328-356). The values at the two read cursors are kept in registers (vs = d[s], vr = d[r]) and re-read only
    // when a cursor moves: one LDS round trip per pick on the chain instead of one per access.
    {
        int s = 0, r = 0;
        uint32_t vs = d[0], vr = 0;
        for (int t = 0; t < n - 1; t++) {
            uint32_t sum = 0;
            for (int i = 0; i < 2; i++) {
                if (s >= n || (r < t && vr < vs)) {
                    sum += vr;
                    d[r * 64] = (uint16_t)t;
                    r++;
                    vr = d[r * 64];                           // (r == t: not stored yet, replaced below)
                    continue;
                }
                sum += vs;
                if (s > t) d[s * 64] = 0;
                s++;
                vs = d[min(s, n - 1) * 64];
            }
            d[t * 64] = (uint16_t)sum;
            if (r == t) vr = sum & 0xFFFFu;
        }
    }
    // phase 2 (:359-385)
    int levelTop = n - 2, depth = 1, i = n, totalNodesAtLevel = 2;
    while (i > 0) {
        int k = levelTop;
        while (k > 0 && (int)d[(k - 1) * 64] >= levelTop) k--;
        const int internalNodesAtLevel = levelTop - k;
        const int leavesAtLevel = totalNodesAtLevel - internalNodesAtLevel;
        for (int j = 0; j < leavesAtLevel; j++) { i--; d[i * 64] = (uint16_t)depth; }
        totalNodesAtLevel = internalNodesAtLevel << 1;
        levelTop = k;
        depth++;
    }
    for (int j = 0; j < n; j++) a.st_len[knz_huf_st(chunk, j)] = (uint8_t)min(255u, (uint32_t)d[j * 64]);
    a.st_maxlen[chunk] = (uint8_t)min(255, depth - 1);
}

// ---- kernel 3: canonical codes, the 4 fragments, the header ------------------------------------------------------------------
// SIZES: everything but the encoding itself: code lengths behind the limiter, the header as it will be written, the fragments' bit counts from
// the per-fragment symbol counts -> unit_bits (what the layout scans need). !SIZES: the encoder; with a.dst_words it places its units at
// their final bit positions, else into the chunk's scratch slot.
template <bool SIZES>
__global__ __launch_bounds__(256) void knz_huf_encode_kernel(HufEncArgs a) {
    __shared__ uint32_t s_out[4][KNZ_FRAG_BYTES / 4];
    __shared__ uint32_t s_freq[256];
    __shared__ uint32_t s_sorted[256];
    __shared__ uint32_t s_data[256];
    __shared__ uint8_t s_len[256];
    __shared__ uint8_t s_alpha[256];
    __shared__ uint16_t s_code[256];
    __shared__ uint32_t s_hdr[KNZ_U0_BYTES / 4];
    __shared__ uint32_t s_fragbits[4];
    __shared__ uint32_t s_cnt[4][16];
    __shared__ uint32_t s_wsum[4];
    __shared__ int s_fallback, s_panic;

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const uint32_t cpb = a.chunks_per_block;
    const uint32_t b = blockIdx.x / cpb, k = blockIdx.x % cpb;
    const uint32_t postLen = a.blk_len[b];
    uint32_t* ubits = a.unit_bits + (size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK;
    const bool direct = !SIZES && a.dst_words != nullptr;
    if (tid < KNZ_UNITS_PER_CHUNK && !direct)
        a.unit_src[(size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK + tid] = tid == 0 ? 0u : (uint32_t)(KNZ_U0_BYTES + (tid - 1) * KNZ_FRAG_BYTES);
    if ((uint64_t)k * KNZ_HUF_CHUNK >= postLen) {
        if (tid < KNZ_UNITS_PER_CHUNK && !direct) ubits[tid] = 0;
        return;
    }
    if (direct && a.total_bits[1] != 0) return;                          // the stream does not fit its buffer: nothing is written
    const uint32_t n = min((uint32_t)KNZ_HUF_CHUNK, postLen - k * KNZ_HUF_CHUNK);
    const uint8_t* src = a.data + a.blk_off[b] + (size_t)k * KNZ_HUF_CHUNK;
    uint8_t* slot = a.scratch + (size_t)blockIdx.x * KNZ_CHUNK_STRIDE;

    // ---- zero LDS staging ---------------------------------------------------------------------------------
    if (!SIZES) for (int i = tid; i < 4 * (KNZ_FRAG_BYTES / 4); i += 256) (&s_out[0][0])[i] = 0;   // (the sizes pass stages no fragment)
    if (tid < KNZ_U0_BYTES / 4) s_hdr[tid] = 0;
    s_freq[tid] = 0; s_len[tid] = 0; s_code[tid] = 0;
    if (tid == 0) { s_panic = 0; s_fallback = 0; }
    __syncthreads();

    if (n < 32) { // HuffmanCodec.go:411-413: raw bytes
        if (tid < (int)n) atomicOr(&s_hdr[tid >> 2], (uint32_t)src[tid] << (24 - 8 * (tid & 3)));
        __syncthreads();
        if (direct) {
            const uint64_t p0 = a.blk_dst_bit[b] + a.chunk_rel[blockIdx.x], p1 = p0 + 8ull * n;
            const uint64_t w0 = p0 >> 5, w1 = (p1 - 1) >> 5;
            for (uint64_t w = w0 + tid; w <= w1; w += 256) {
                const uint32_t sw = knz_bswap32(knz_lds_fetch32(s_hdr, (int64_t)(w << 5) - (int64_t)p0, 8 * (int64_t)n));
                if (w == w0 || w == w1) atomicOr(&a.dst_words[w], sw); else a.dst_words[w] = sw;
            }
            return;
        }
        if (!SIZES && tid < 8) ((uint32_t*)slot)[tid] = knz_bswap32(s_hdr[tid]);
        if (tid < KNZ_UNITS_PER_CHUNK) ubits[tid] = (tid == 0) ? 8u * n : 0u;
        return;
    }

    // ---- statistics of kernels 1 and 2: rank order -> symbol order ---------------------------------------------------------
    const int count = (int)a.st_count[blockIdx.x];
    if (tid < count) {
        const uint32_t sym = a.st_sym[knz_huf_st(blockIdx.x, tid)];
        s_sorted[tid] = sym;
        s_freq[sym] = a.st_freq[knz_huf_st(blockIdx.x, tid)];
        s_len[sym] = a.st_len[knz_huf_st(blockIdx.x, tid)];
    }
    const int maxLen0 = (int)a.st_maxlen[blockIdx.x];                   // before the limiter (every thread reads its own copy)
    __syncthreads();
    // alphabet = present symbols in increasing order (HuffmanCodec.go:137-146)
    const bool present = s_freq[tid] > 0;
    const uint64_t bal = wave_ballot(present);
    if (lane == 0) s_wsum[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; w++) before += s_wsum[w];
    const uint32_t myIdx = before + (uint32_t)__popcll(bal & (((uint64_t)1 << lane) - 1));
    if (present) s_alpha[myIdx] = (uint8_t)tid;
    __syncthreads();

    // ---- 12-bit limiter (limitCodeLengths :216-297): rare, serial on one thread ------------------------------------------------
    if (maxLen0 > KNZ_HUF_MAXLEN) {
        if (tid == 0) {
            int panic = 0;
            int maxLen = knz_huf_limit(s_alpha, count, s_freq, s_len, s_sorted, s_data, &panic, &s_out[0][0]);
            if (maxLen > KNZ_HUF_MAXLEN) {
                s_fallback = 1;
                for (int i = 0; i < count; i++) s_len[s_alpha[i]] = 8;
            }
            s_panic = panic;
        }
        __syncthreads();
        for (int i = tid; i < 1024; i += 256) s_out[0][i] = 0;   // the limiter used the head of the staging area as scratch
        __syncthreads();
    }
    if (s_panic) { // Go would panic (index out of range) -> encodingTask recovers it as ERR_PROCESS_BLOCK
        if (tid == 0) a.blk_status[b] = 13;
        if (tid < KNZ_UNITS_PER_CHUNK && !direct) ubits[tid] = 0;
        return;
    }

    // ---- canonical codes (generateCanonicalCodes :37-77): handed out by (length, symbol), one ballot per length;
    //      fallback: alphabet index on 8 bits (:179-184) -----------------------------------------------------------------------
    const uint32_t myLen = s_len[tid];
    {
        uint32_t myBefore = 0;
        const uint64_t below = ((uint64_t)1 << lane) - 1;
        for (uint32_t L = 1; L <= KNZ_HUF_MAXLEN; L++) {
            const uint64_t m = wave_ballot(myLen == L);
            if (lane == 0) s_cnt[wave][L] = (uint32_t)__popcll(m);
            if (myLen == L) myBefore = (uint32_t)__popcll(m & below);
        }
        __syncthreads();
        uint32_t full = 0, myC = 0;
        for (uint32_t L = 1; L <= KNZ_HUF_MAXLEN; L++) {
            const uint32_t c0 = s_cnt[0][L], c1 = s_cnt[1][L], c2 = s_cnt[2][L], c3 = s_cnt[3][L];
            if (L == myLen) myC = full + (((wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u) + myBefore) << (KNZ_HUF_MAXLEN - L));
            full += (c0 + c1 + c2 + c3) << (KNZ_HUF_MAXLEN - L);
        }
        if (present) {
            const uint32_t code = s_fallback ? myIdx : (myC >> (KNZ_HUF_MAXLEN - myLen));
            s_code[tid] = (uint16_t)((myLen << 12) | (code & 0x0FFF));
        }
    }
    __syncthreads();

    // ---- fragments: wave j encodes symbols [j*F, (j+1)*F) (encodeChunk :435-492) -------------------------------
    const uint32_t F = n >> 2;
    if (SIZES) {
        // bit count of fragment `wave` = sum over the symbols of (count in the fragment) x (code length)
        uint32_t nb = 0;
        if (count > 1) {
            const uint16_t* fh = a.st_fhist + ((size_t)blockIdx.x * 4 + wave) * 256;
#pragma unroll
            for (int q = 0; q < 4; q++) nb += (uint32_t)fh[64 * q + lane] * (uint32_t)(s_code[64 * q + lane] >> 12);
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) nb += wave_shfl(nb, lane ^ d);
        }
        if (lane == 0) s_fragbits[wave] = nb;
    } else if (count > 1) {
        const uint32_t S = (F + 63) >> 6;                 // symbols per lane
        const uint32_t first = min(F, (uint32_t)lane * S);
        const uint32_t last = min(F, first + S);
        const uint8_t* fsrc = src + (size_t)wave * F;
        // full chunks: 64 symbols per lane, fetched as 4 x 16 bytes up front (one cache line per lane) and walked in registers
        const bool fast = S == 64 && F == 4096 && ((((uintptr_t)fsrc) & 15) == 0);
        uint32_t wd[16];
        if (fast) {
            const uint4* q = (const uint4*)(fsrc + first);
#pragma unroll
            for (int u = 0; u < 4; u++) { const uint4 x = q[u]; wd[4 * u] = x.x; wd[4 * u + 1] = x.y; wd[4 * u + 2] = x.z; wd[4 * u + 3] = x.w; }
        }
        uint32_t nbits = 0;
        if (fast) {
#pragma unroll
            for (int i = 0; i < 64; i++) nbits += s_code[(wd[i >> 2] >> (8 * (i & 3))) & 0xFF] >> 12;
        } else {
            for (uint32_t i = first; i < last; i++) nbits += s_code[fsrc[i]] >> 12;
        }
        const uint32_t incl = wave_scan_incl(nbits);
        const uint32_t start = incl - nbits;
        const uint32_t total = wave_bcast(incl, 63);
        if (lane == 0) s_fragbits[wave] = total;
        uint32_t* out = s_out[wave];
        uint64_t acc = 0;
        uint32_t cnt = start & 31;
        uint32_t w = start >> 5;
        bool firstWord = true;
        auto put = [&](uint32_t c) {
            const uint32_t L = c >> 12;
            acc = (acc << L) | (c & 0x0FFF);
            cnt += L;
            if (cnt >= 32) {
                const uint32_t word = (uint32_t)(acc >> (cnt - 32));
                if (firstWord) { atomicOr(&out[w], word); firstWord = false; }
                else out[w] = word;
                w++;
                cnt -= 32;
            }
        };
        if (fast) {
#pragma unroll
            for (int i = 0; i < 64; i++) put(s_code[(wd[i >> 2] >> (8 * (i & 3))) & 0xFF]);
        } else {
            for (uint32_t i = first; i < last; i++) put(s_code[fsrc[i]]);
        }
        if (cnt > 0 && nbits > 0) atomicOr(&out[w], (uint32_t)(acc << (32 - cnt)) );
    } else if (lane == 0) {
        s_fragbits[wave] = 0;
    }

    // ---- unit 0: alphabet (EntropyUtils.go:46-60), code length deltas as signed Exp-Golomb codes (updateFrequencies
    //      :148,186-210), one thread per alphabet entry: bit positions by a prefix sum, bits OR-ed into place ----------------------
    uint32_t hdrBase;
    if (count == 256) hdrBase = 2;                                      // '0' '0': full alphabet
    else {
        const uint32_t lastMask = (uint32_t)s_alpha[count - 1] >> 3;
        hdrBase = 6 + 8 * (lastMask + 1);
        if (tid == 0) atomicOr(&s_hdr[0], (1u << 31) | (lastMask << 26));
        if ((uint32_t)tid <= lastMask) {
            uint32_t mask = 0;
            for (int bit = 0; bit < 8; bit++) mask |= (s_freq[8 * tid + bit] ? 1u : 0u) << bit;
            const uint32_t p = 6 + 8 * (uint32_t)tid;                   // 8 bits at bit p
            const uint32_t wi = p >> 5, off = p & 31;
            if (off <= 24) atomicOr(&s_hdr[wi], mask << (24 - off));
            else { atomicOr(&s_hdr[wi], mask >> (off - 24)); atomicOr(&s_hdr[wi + 1], mask << (56 - off)); }
        }
    }
    uint32_t e = 0;
    if (tid < count) {
        const int cur = s_len[s_alpha[tid]];
        const int prev = tid > 0 ? (int)s_len[s_alpha[tid - 1]] : 2;
        e = knz_expg_signed((int)(int8_t)(uint8_t)(cur - prev));
    }
    const uint32_t elen = e >> 9;
    const uint32_t eincl = wave_scan_incl(elen);
    __syncthreads();                                                    // s_wsum is free again
    if (lane == 63) s_wsum[wave] = eincl;
    __syncthreads();
    uint32_t epos = hdrBase + eincl - elen;
    for (int w = 0; w < wave; w++) epos += s_wsum[w];
    if (elen) {
        const uint32_t bitsv = e & 0x1FF;
        const uint32_t wi = epos >> 5, off = epos & 31;
        if (off + elen <= 32) atomicOr(&s_hdr[wi], bitsv << (32 - off - elen));
        else { atomicOr(&s_hdr[wi], bitsv >> (off + elen - 32)); atomicOr(&s_hdr[wi + 1], bitsv << (64 - off - elen)); }
    }
    const uint32_t hdrBits = hdrBase + s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
    __syncthreads();
    if (tid == 0) {                                                     // 4 varint fragment sizes (encodeChunk :494-497)
        KnzBitWriter bw;
        bw.init(s_hdr);
        bw.pos = hdrBits;
        if (count > 1) for (int j = 0; j < 4; j++) knz_put_varint(bw, s_fragbits[j]);
        s_data[0] = bw.pos;
    }
    // tail bytes go behind fragment 3 (encodeChunk :505-510)
    if (tid == 64 && count > 1) {
        if (SIZES) s_data[1] = s_fragbits[3] + 8 * (n - 4 * F);
        else {
            KnzBitWriter bw;
            bw.init(s_out[3]);
            bw.pos = s_fragbits[3];
            for (uint32_t i = 4 * F; i < n; i++) bw.put(src[i], 8);
            s_data[1] = bw.pos;
        }
    }
    __syncthreads();

    const uint32_t u0bits = s_data[0];
    const uint32_t u4bits = count > 1 ? s_data[1] : 0;
    if (SIZES) {
        if (tid == 0) {
            ubits[0] = u0bits;
            ubits[1] = s_fragbits[0]; ubits[2] = s_fragbits[1]; ubits[3] = s_fragbits[2];
            ubits[4] = u4bits;
        }
        return;
    }
    if (direct) {
        // ---- the units go straight to their final bit positions: every thread owns destination words and ORs in what each of the five units
        //      contributes (the funnel shift of knz_gather_kernel, fed from LDS); only the chunk's first and last word are shared with neighbours
        //      (zeroed by knz_layout_stream_kernel, which has run: it only needs the bit counts) ---------------------------------------------
        uint64_t ustart[KNZ_UNITS_PER_CHUNK + 1];
        ustart[0] = a.blk_dst_bit[b] + a.chunk_rel[blockIdx.x];
        ustart[1] = ustart[0] + u0bits; ustart[2] = ustart[1] + s_fragbits[0]; ustart[3] = ustart[2] + s_fragbits[1];
        ustart[4] = ustart[3] + s_fragbits[2]; ustart[5] = ustart[4] + u4bits;
        if (tid == 0 && (ubits[0] != u0bits || ubits[1] != s_fragbits[0] || ubits[2] != s_fragbits[1] || ubits[3] != s_fragbits[2] || ubits[4] != u4bits))
            a.blk_status[b] = KNZ_ERR_UNKNOWN;                             // the sizes pass and the encoder disagree: must not happen
        const uint64_t p0 = ustart[0], p1 = ustart[KNZ_UNITS_PER_CHUNK];
        if (p1 == p0) return;
        const uint64_t w0 = p0 >> 5, w1 = (p1 - 1) >> 5;
        for (uint64_t w = w0 + tid; w <= w1; w += 256) {
            const int64_t wbit = (int64_t)(w << 5);
            uint32_t v = knz_lds_fetch32(s_hdr, wbit - (int64_t)ustart[0], (int64_t)u0bits);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t us = (int64_t)ustart[j + 1], ue = (int64_t)ustart[j + 2];
                if (ue <= wbit || us >= wbit + 32 || ue == us) continue;
                v |= knz_lds_fetch32(s_out[j], wbit - us, ue - us);
            }
            const uint32_t sw = knz_bswap32(v);
            if (w == w0 || w == w1) atomicOr(&a.dst_words[w], sw);
            else a.dst_words[w] = sw;
        }
        return;
    }
    // ---- copy the units to the scratch slot (byte-swapped BE words) -------------------------------------------------
    {
        uint32_t* g = (uint32_t*)slot;
        for (uint32_t i = tid; i < ((u0bits + 31) >> 5); i += 256) g[i] = knz_bswap32(s_hdr[i]);
        for (int j = 0; j < 4; j++) {
            uint32_t bits = (j == 3) ? u4bits : s_fragbits[j];
            uint32_t* gf = (uint32_t*)(slot + KNZ_U0_BYTES + (size_t)j * KNZ_FRAG_BYTES);
            for (uint32_t i = tid; i < ((bits + 31) >> 5); i += 256) gf[i] = knz_bswap32(s_out[j][i]);
        }
    }
    if (tid == 0) {
        ubits[0] = u0bits;
        ubits[1] = s_fragbits[0]; ubits[2] = s_fragbits[1]; ubits[3] = s_fragbits[2];
        ubits[4] = u4bits;
    }
}
