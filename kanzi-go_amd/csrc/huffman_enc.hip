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
};

__global__ __launch_bounds__(256) void knz_huf_encode_kernel(HufEncArgs a) {
    __shared__ uint32_t s_out[4][KNZ_FRAG_BYTES / 4];
    __shared__ uint32_t s_hist[4][256];
    __shared__ uint32_t s_freq[256];
    __shared__ uint32_t s_sorted[256];
    __shared__ uint32_t s_data[256];
    __shared__ uint8_t s_len[256];
    __shared__ uint8_t s_alpha[256];
    __shared__ uint16_t s_code[256];
    __shared__ uint32_t s_hdr[KNZ_U0_BYTES / 4];
    __shared__ uint32_t s_fragbits[4];
    __shared__ int s_maxLen, s_fallback, s_panic, s_limited;

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const uint32_t cpb = a.chunks_per_block;
    const uint32_t b = blockIdx.x / cpb, k = blockIdx.x % cpb;
    const uint32_t postLen = a.blk_len[b];
    uint32_t* ubits = a.unit_bits + (size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK;
    if (tid < KNZ_UNITS_PER_CHUNK)
        a.unit_src[(size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK + tid] = tid == 0 ? 0u : (uint32_t)(KNZ_U0_BYTES + (tid - 1) * KNZ_FRAG_BYTES);
    if ((uint64_t)k * KNZ_HUF_CHUNK >= postLen) {
        if (tid < KNZ_UNITS_PER_CHUNK) ubits[tid] = 0;
        return;
    }
    const uint32_t n = min((uint32_t)KNZ_HUF_CHUNK, postLen - k * KNZ_HUF_CHUNK);
    const uint8_t* src = a.data + a.blk_off[b] + (size_t)k * KNZ_HUF_CHUNK;
    uint8_t* slot = a.scratch + (size_t)blockIdx.x * KNZ_CHUNK_STRIDE;

    // ---- zero LDS staging ---------------------------------------------------------------------------------
    for (int i = tid; i < 4 * (KNZ_FRAG_BYTES / 4); i += 256) (&s_out[0][0])[i] = 0;
    for (int i = tid; i < 4 * 256; i += 256) (&s_hist[0][0])[i] = 0;
    if (tid < KNZ_U0_BYTES / 4) s_hdr[tid] = 0;
    if (tid == 0) { s_panic = 0; s_fallback = 0; s_maxLen = 0; s_limited = 0; }
    __syncthreads();

    if (n < 32) { // HuffmanCodec.go:411-413: raw bytes
        if (tid < (int)n) atomicOr(&s_hdr[tid >> 2], (uint32_t)src[tid] << (24 - 8 * (tid & 3)));
        __syncthreads();
        if (tid < 8) ((uint32_t*)slot)[tid] = knz_bswap32(s_hdr[tid]);
        if (tid < KNZ_UNITS_PER_CHUNK) ubits[tid] = (tid == 0) ? 8u * n : 0u;
        return;
    }

    // ---- histogram (Global.go:226-251): coalesced 16 B per lane, per-wave private counters --------------------
    knz_histogram_256t(src, n, s_hist, tid);
    __syncthreads();
    const uint32_t myFreq = s_hist[0][tid] + s_hist[1][tid] + s_hist[2][tid] + s_hist[3][tid];
    s_freq[tid] = myFreq;
    s_len[tid] = 0;
    s_code[tid] = 0;
    // alphabet = present symbols in increasing order (HuffmanCodec.go:137-146)
    const bool present = myFreq > 0;
    const uint64_t bal = wave_ballot(present);
    if (lane == 0) s_data[wave] = (uint32_t)__popcll(bal);   // per-wave counts, reused below
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; w++) before += s_data[w];
    const int count = (int)(s_data[0] + s_data[1] + s_data[2] + s_data[3]);
    const uint32_t myIdx = before + (uint32_t)__popcll(bal & (((uint64_t)1 << lane) - 1));
    __syncthreads();               // s_data is about to be reused
    if (present) s_alpha[myIdx] = (uint8_t)tid;

    // ---- sort keys (freq<<8|sym) by rank counting (sort.Ints in computeCodeLengths :303) -----------------------
    const uint32_t myKey = present ? ((myFreq << 8) | (uint32_t)tid) : 0xFFFFFFFFu;
    if (count > 1) {
        uint32_t rank = 0;
        for (int u = 0; u < 256; u++) {
            uint32_t f = s_freq[u];
            uint32_t key = f ? ((f << 8) | (uint32_t)u) : 0xFFFFFFFFu;
            rank += key < myKey ? 1u : 0u;
        }
        if (present) s_sorted[rank] = myKey;
    }
    __syncthreads();

    // ---- code lengths: serial section on one thread (Moffat-Katajainen + limiter) ------------------------------
    if (tid == 0) {
        if (count == 1) {
            s_len[s_alpha[0]] = 1;      // HuffmanCodec.go:156-158
            s_maxLen = 1;
        } else {
            for (int i = 0; i < count; i++) { s_data[i] = s_sorted[i] >> 8; s_sorted[i] &= 0xFF; }
            int maxLen = knz_huf_lengths_inplace(s_data, count);
            for (int i = 0; i < count; i++) s_len[s_sorted[i]] = (uint8_t)s_data[i];
            int panic = 0;
            if (maxLen > KNZ_HUF_MAXLEN) { s_limited = 1; maxLen = knz_huf_limit(s_alpha, count, s_freq, s_len, s_sorted, s_data, &panic, &s_out[0][0]); }
            if (maxLen > KNZ_HUF_MAXLEN) {
                s_fallback = 1;
                for (int i = 0; i < count; i++) s_len[s_alpha[i]] = 8;
            }
            s_maxLen = maxLen;
            s_panic = panic;
        }
    }
    __syncthreads();
    if (s_limited) {   // the limiter used the head of the fragment staging area as scratch: zero it again
        for (int i = tid; i < 1024; i += 256) s_out[0][i] = 0;
        __syncthreads();
    }
    if (s_panic) { // Go would panic (index out of range) -> encodingTask recovers it as ERR_PROCESS_BLOCK
        if (tid == 0) a.blk_status[b] = 13;
        if (tid < KNZ_UNITS_PER_CHUNK) ubits[tid] = 0;
        return;
    }

    // ---- canonical codes (generateCanonicalCodes :37-77): code = (sum of 2^(12-len) over symbols that sort
    //      before (len,sym)) >> (12-len); fallback: alphabet index on 8 bits (:179-184) ---------------------------
    const uint32_t myLen = s_len[tid];
    if (present) {
        uint32_t code;
        if (s_fallback) code = myIdx;
        else {
            uint32_t acc = 0;
            for (int u = 0; u < 256; u++) {
                uint32_t lu = s_len[u];
                bool beforeMe = lu != 0 && (lu < myLen || (lu == myLen && u < tid));
                acc += beforeMe ? (1u << (KNZ_HUF_MAXLEN - lu)) : 0u;
            }
            code = acc >> (KNZ_HUF_MAXLEN - myLen);
        }
        s_code[tid] = (uint16_t)((myLen << 12) | (code & 0x0FFF));
    }
    __syncthreads();

    // ---- fragments: wave j encodes symbols [j*F, (j+1)*F) (encodeChunk :435-492) -------------------------------
    const uint32_t F = n >> 2;
    if (count > 1) {
        const uint32_t S = (F + 63) >> 6;                 // symbols per lane
        const uint32_t first = min(F, (uint32_t)lane * S);
        const uint32_t last = min(F, first + S);
        const uint8_t* fsrc = src + (size_t)wave * F;
        uint32_t nbits = 0;
        for (uint32_t i = first; i < last; i++) nbits += s_code[fsrc[i]] >> 12;
        const uint32_t incl = wave_scan_incl(nbits);
        const uint32_t start = incl - nbits;
        const uint32_t total = wave_bcast(incl, 63);
        if (lane == 0) s_fragbits[wave] = total;
        uint32_t* out = s_out[wave];
        uint64_t acc = 0;
        uint32_t cnt = start & 31;
        uint32_t w = start >> 5;
        bool firstWord = true;
        for (uint32_t i = first; i < last; i++) {
            uint32_t c = s_code[fsrc[i]];
            uint32_t L = c >> 12;
            acc = (acc << L) | (c & 0x0FFF);
            cnt += L;
            if (cnt >= 32) {
                uint32_t word = (uint32_t)(acc >> (cnt - 32));
                if (firstWord) { atomicOr(&out[w], word); firstWord = false; }
                else out[w] = word;
                w++;
                cnt -= 32;
            }
        }
        if (cnt > 0 && nbits > 0) atomicOr(&out[w], (uint32_t)(acc << (32 - cnt)) );
    } else if (lane == 0) {
        s_fragbits[wave] = 0;
    }
    __syncthreads();

    // ---- unit 0: alphabet, code length deltas, fragment sizes (updateFrequencies :148,186-210, encodeChunk :494-497)
    if (tid == 0) {
        KnzBitWriter bw;
        bw.init(s_hdr);
        if (count == 256) { bw.put(0, 1); bw.put(0, 1); }     // EntropyUtils.go:46-51
        else {
            bw.put(1, 1);
            int lastMask = s_alpha[count - 1] >> 3;
            bw.put((uint32_t)lastMask, 5);
            for (int m = 0; m <= lastMask; m++) {
                uint32_t mask = 0;
                for (int bit = 0; bit < 8; bit++) mask |= (s_freq[8 * m + bit] ? 1u : 0u) << bit;
                bw.put(mask, 8);
            }
        }
        int prev = 2;
        for (int i = 0; i < count; i++) {
            int cur = s_len[s_alpha[i]];
            uint32_t e = knz_expg_signed((int)(int8_t)(uint8_t)(cur - prev));
            bw.put(e & 0x1FF, e >> 9);
            prev = cur;
        }
        if (count > 1) for (int j = 0; j < 4; j++) knz_put_varint(bw, s_fragbits[j]);
        s_data[0] = bw.pos;
    }
    // tail bytes go behind fragment 3 (encodeChunk :505-510)
    if (tid == 64 && count > 1) {
        KnzBitWriter bw;
        bw.init(s_out[3]);
        bw.pos = s_fragbits[3];
        for (uint32_t i = 4 * F; i < n; i++) bw.put(src[i], 8);
        s_data[1] = bw.pos;
    }
    __syncthreads();

    // ---- copy the units to the scratch slot (byte-swapped BE words) -------------------------------------------------
    const uint32_t u0bits = s_data[0];
    const uint32_t u4bits = count > 1 ? s_data[1] : 0;
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
