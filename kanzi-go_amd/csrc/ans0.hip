// Static rANS order 0 of kanzi bitstream v6 on gfx950 (16 KiB chunks, log range 12, 4 interleaved states feeding ONE
// reverse byte stream). Replaces ANSRangeEncoder.Write / rebuildStatistics / updateFrequencies / encodeHeader /
// encodeChunk / encodeSymbol and encSymbol.reset (v2/entropy/ANSRangeCodec.go:274,408,171,216,331,313,446),
// EntropyUtils.NormalizeFrequencies (v2/entropy/EntropyUtils.go:123-260) and the decoder
// (decodeHeader :605, decodeChunkV2 :860, decodeSymbol :846).
//
// Encode = two kernels:
//   knz_ans0_stats_kernel  : one 256-thread workgroup per chunk: histogram, normalisation to 4096, symbol parameters
//                            (freq, bias, Alverson reciprocal) into a 2 KiB table, header bits (unit 0).
//   knz_ans0_encode_kernel : one LANE per rANS state: 4 adjacent lanes own a chunk, 16 chunks per wave64. The four
//                            states of a chunk advance in lock-step, so the position of every 16-bit renormalisation
//                            word in the single shared (descending) stream is a wave ballot + popcount: no second pass.
//                            Words are written right-aligned into the chunk's scratch slot (unit 2), final states and
//                            the byte count form unit 1.
// The chain itself is the format's serial dependency (4096 steps per state); parallelism = 4 x chunks.
#include "bits.h"

struct Ans0Args {
    const uint8_t* data;
    const uint64_t* blk_off;
    const uint32_t* blk_len;
    uint32_t chunks_per_block;
    uint8_t* scratch;              // [slots * KNZ_ANS_SLOT]
    uint32_t* unit_bits;           // [slots * 5]
    uint32_t* unit_src;            // [slots * 5]
    uint2* tab;                    // [slots * 256] {freq | bias<<12 | (invShift-32)<<25, invFreq}
    uint32_t* chunk_info;          // [slots] alphabet size (0 = chunk absent / raw)
    int32_t* blk_status;
};

__global__ __launch_bounds__(256) void knz_ans0_stats_kernel(Ans0Args a) {
    __shared__ uint32_t s_hist[4][256];
    __shared__ int s_f[256];
    __shared__ int s_alpha[256];
    __shared__ uint32_t s_hdr[KNZ_U0_BYTES / 4];
    __shared__ uint32_t s_wsum[4];
    __shared__ int s_asize, s_panic;
    __shared__ uint32_t s_hbits;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cpb = a.chunks_per_block;
    const uint32_t b = blockIdx.x / cpb, k = blockIdx.x % cpb;
    const uint32_t postLen = a.blk_len[b];
    uint32_t* ubits = a.unit_bits + (size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK;
    uint32_t* usrc = a.unit_src + (size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK;
    if (tid < KNZ_UNITS_PER_CHUNK) { ubits[tid] = 0; usrc[tid] = tid == 1 ? KNZ_ANS_U1_OFF : (tid == 2 ? KNZ_ANS_PAY_OFF : 0u); }
    if (tid == 0) a.chunk_info[blockIdx.x] = 0;
    if ((uint64_t)k * KNZ_ANS_CHUNK >= postLen) return;
    const uint32_t n = min((uint32_t)KNZ_ANS_CHUNK, postLen - k * KNZ_ANS_CHUNK);
    const uint8_t* src = a.data + a.blk_off[b] + (size_t)k * KNZ_ANS_CHUNK;
    uint8_t* slot = a.scratch + (size_t)blockIdx.x * KNZ_ANS_SLOT;

    for (int i = tid; i < 4 * 256; i += 256) (&s_hist[0][0])[i] = 0;
    if (tid < KNZ_U0_BYTES / 4) s_hdr[tid] = 0;
    if (tid == 0) { s_panic = 0; s_asize = 0; }
    __syncthreads();

    if (postLen <= 32) {   // ANSRangeEncoder.Write :279-282: whole input raw
        if (tid < (int)n) atomicOr(&s_hdr[tid >> 2], (uint32_t)src[tid] << (24 - 8 * (tid & 3)));
        __syncthreads();
        if (tid < 8) ((uint32_t*)slot)[tid] = knz_bswap32(s_hdr[tid]);
        if (tid == 0) ubits[0] = 8u * n;
        return;
    }

    knz_histogram_256t(src, n, s_hist, tid);          // order 0 counts every byte of the chunk, tail included (:412)
    __syncthreads();
    s_f[tid] = (int)(s_hist[0][tid] + s_hist[1][tid] + s_hist[2][tid] + s_hist[3][tid]);
    __syncthreads();
    if (tid == 0) {
        int panic = 0;
        s_asize = knz_normalize_freqs(s_f, 256, s_alpha, (int)n, 1 << 12, &panic);   // updateFrequencies :185
        s_panic = panic;
    }
    __syncthreads();
    const int asize = s_asize;

    // cumulated frequencies in symbol order + encSymbol.reset (:446-468)
    const uint32_t f = (uint32_t)s_f[tid];
    const uint32_t incl = wave_scan_incl(f);
    if (lane == 63) s_wsum[wave] = incl;
    __syncthreads();
    uint32_t cum = incl - f;
    for (int w = 0; w < wave; w++) cum += s_wsum[w];
    {
        uint32_t fr = f < 4095u ? f : 4095u;              // freq = min(freq, (1<<logRange)-1)
        uint32_t w0 = 0, w1 = 0;
        if (f != 0) {
            uint32_t bias, invFreq, sh;
            if (fr < 2) { invFreq = 0xFFFFFFFFu; sh = 0; bias = cum + 4095u; }
            else {
                uint32_t shift = 0;
                while (fr > (1u << shift)) shift++;
                invFreq = (uint32_t)(((((uint64_t)1) << (shift + 31)) + (uint64_t)(fr - 1)) / (uint64_t)fr);
                sh = shift - 1;                            // invShift = 32 + shift - 1
                bias = cum;
            }
            w0 = fr | (bias << 12) | (sh << 25);
            w1 = invFreq;
        }
        uint2 e; e.x = w0; e.y = w1;
        a.tab[(size_t)blockIdx.x * 256 + tid] = e;
    }

    // header: (lr-8):3, alphabet, frequencies by groups (encodeHeader :216-270)
    if (tid == 0) {
        KnzBitWriter bw;
        bw.init(s_hdr);
        bw.put(12 - 8, 3);
        if (asize == 256) { bw.put(0, 1); bw.put(0, 1); }
        else if (asize == 0) { bw.put(0, 1); bw.put(1, 1); }
        else {
            bw.put(1, 1);
            int lastMask = s_alpha[asize - 1] >> 3;
            bw.put((uint32_t)lastMask, 5);
            uint32_t masks[32];
            for (int m = 0; m < 32; m++) masks[m] = 0;
            for (int i = 0; i < asize; i++) masks[s_alpha[i] >> 3] |= 1u << (s_alpha[i] & 7);
            for (int m = 0; m <= lastMask; m++) bw.put(masks[m], 8);
        }
        if (asize > 1) {
            const int chk = asize < 64 ? 6 : 8;
            const uint32_t llr = 4;                        // smallest llr with 1<<llr > 12
            for (int i = 1; i < asize; i += chk) {
                int mx = s_f[s_alpha[i]] - 1;
                const int endj = min(i + chk, asize);
                for (int j = i + 1; j < endj; j++) { int v = s_f[s_alpha[j]] - 1; if (v > mx) mx = v; }
                uint32_t logMax = 0;
                while ((1 << logMax) <= mx) logMax++;
                bw.put(logMax, llr);
                if (logMax == 0) continue;
                for (int j = i; j < endj; j++) bw.put((uint32_t)(s_f[s_alpha[j]] - 1), logMax);
            }
        }
        s_hbits = bw.pos;
    }
    __syncthreads();
    if (s_panic) { if (tid == 0) a.blk_status[b] = KNZ_ERR_PROCESS_BLOCK; return; }
    const uint32_t hb = s_hbits;
    for (uint32_t i = tid; i < ((hb + 31) >> 5); i += 256) ((uint32_t*)slot)[i] = knz_bswap32(s_hdr[i]);
    if (tid == 0) { ubits[0] = hb; a.chunk_info[blockIdx.x] = (uint32_t)asize; }
}

// 128 threads = 32 chunks per workgroup, 2 KiB symbol table per chunk in LDS (64 KiB).
#define KNZ_ANS0_CHUNKS_PER_WG 32

__global__ __launch_bounds__(128) void knz_ans0_encode_kernel(Ans0Args a, uint32_t nslots) {
    __shared__ uint2 s_tab[KNZ_ANS0_CHUNKS_PER_WG][256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int g = tid >> 2;                  // chunk inside the workgroup
    const int c = tid & 3;                   // state index: st0..st3
    const uint32_t slotId = blockIdx.x * KNZ_ANS0_CHUNKS_PER_WG + g;
    const uint32_t cpb = a.chunks_per_block;
    bool live = slotId < nslots;
    uint32_t n = 0;
    const uint8_t* src = a.data;
    if (live) {
        const uint32_t b = slotId / cpb, k = slotId % cpb;
        const uint32_t postLen = a.blk_len[b];
        if ((uint64_t)k * KNZ_ANS_CHUNK >= postLen || postLen <= 32 || a.chunk_info[slotId] <= 1 || a.blk_status[b] != 0) live = false;
        else {
            n = min((uint32_t)KNZ_ANS_CHUNK, postLen - k * KNZ_ANS_CHUNK);
            src = a.data + a.blk_off[b] + (size_t)k * KNZ_ANS_CHUNK;
        }
    }
    // stage the symbol tables (coalesced: 128 threads x 8 B)
    for (int cg = 0; cg < KNZ_ANS0_CHUNKS_PER_WG; cg++) {
        const uint32_t sid = blockIdx.x * KNZ_ANS0_CHUNKS_PER_WG + cg;
        if (sid < nslots) { s_tab[cg][tid] = a.tab[(size_t)sid * 256 + tid]; s_tab[cg][tid + 128] = a.tab[(size_t)sid * 256 + tid + 128]; }
    }
    __syncthreads();

    uint8_t* slot = a.scratch + (size_t)slotId * KNZ_ANS_SLOT;
    uint8_t* payEnd = slot + KNZ_ANS_PAY_OFF + KNZ_ANS_PAY_CAP;      // renormalisation words grow downwards from here
    const uint32_t end4 = n & ~3u;
    const uint32_t T = live ? (end4 >> 2) : 0;                        // encodeChunk :347-352: i = end4-1 .. 3 step -4
    uint32_t maxT = T;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = wave_shfl(maxT, lane ^ d); maxT = o > maxT ? o : maxT; }
    uint32_t st = 1u << 15;                                            // _ANS_TOP
    uint32_t cnt = 0;                                                  // words emitted so far by this chunk
    const int gshift = (lane >> 2) << 2;
    for (uint32_t t = 0; t < maxT; t++) {
        const bool act = t < T;
        uint32_t x = 0, sym = 0;
        uint2 e; e.x = 0; e.y = 0;
        if (act) {
            const uint32_t i = end4 - 1 - 4 * t;                       // st0 <- block[i], st1 <- block[i-1], ...
            const uint32_t w = *(const uint32_t*)(src + i - 3);        // bytes i-3..i (4-byte aligned)
            sym = (w >> (8 * (3 - c))) & 0xFF;
            e = s_tab[g][sym];
            x = st >= ((e.x & 0xFFFu) << 19) ? 1u : 0u;                // xMax = ((ANS_TOP>>12)<<16)*freq
        }
        const uint64_t bal = wave_ballot(x != 0);
        const uint32_t gb = (uint32_t)(bal >> gshift) & 0xFu;          // the chunk's 4 emit flags, bit c = state c
        if (x) {
            const uint32_t r = cnt + (uint32_t)__popc(gb & ((1u << c) - 1u));   // st0 writes first (:313-329 order)
            uint8_t* p = payEnd - 2 * (r + 1);
            p[0] = (uint8_t)(st >> 8);
            p[1] = (uint8_t)st;
            st >>= 16;
        }
        cnt += (uint32_t)__popc(gb);
        if (act) {
            const uint32_t freq = e.x & 0xFFFu, bias = (e.x >> 12) & 0x1FFFu, sh = (e.x >> 25) & 0xFu;
            const uint32_t q = (uint32_t)(((uint64_t)st * e.y) >> (32 + sh));
            st = st + bias + q * (4096u - freq);
        }
    }
    // unit 1: varint(byte count) + 4 final states (:393-400); unit 2: words + (n & 3) tail bytes (:339-342)
    const uint32_t s1 = wave_shfl(st, (lane & ~3) + 1), s2 = wave_shfl(st, (lane & ~3) + 2), s3 = wave_shfl(st, (lane & ~3) + 3);
    if (live && c == 0) {
        const uint32_t tail = n & 3;
        for (uint32_t i = 0; i < tail; i++) payEnd[i] = src[end4 + i];
        const uint32_t nbytes = 2 * cnt + tail;
        uint32_t w[8];
        for (int i = 0; i < 8; i++) w[i] = 0;
        KnzBitWriter bw;
        bw.init(w);
        knz_put_varint(bw, nbytes);
        bw.put(st, 32); bw.put(s1, 32); bw.put(s2, 32); bw.put(s3, 32);
        uint32_t* u1 = (uint32_t*)(slot + KNZ_ANS_U1_OFF);
        for (int i = 0; i < 8; i++) u1[i] = knz_bswap32(w[i]);
        uint32_t* ubits = a.unit_bits + (size_t)slotId * KNZ_UNITS_PER_CHUNK;
        uint32_t* usrc = a.unit_src + (size_t)slotId * KNZ_UNITS_PER_CHUNK;
        ubits[1] = bw.pos;
        ubits[2] = 8 * nbytes;
        usrc[2] = KNZ_ANS_PAY_OFF + KNZ_ANS_PAY_CAP - 2 * cnt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Decoder. One wave64 per 8 chunks: per chunk the header is parsed by one lane, the slot->symbol table f2s (4 KiB) and the
// symbol table are built by all lanes, then 32 lanes run the 8 x 4 states in lock-step. The single shared read cursor of
// a chunk (decodeChunkV2 :906-949: states refill in the order st3, st2, st1, st0) is again a ballot + popcount.
struct Ans0DecArgs {
    const uint8_t* stream; uint64_t nbytes;
    const uint32_t* blk_pre_len;
    const uint8_t* blk_mode;
    const uint64_t* chunk_bit;
    const uint64_t* blk_out_off;
    uint32_t chunks_per_block;
    uint32_t nslots;
    uint8_t* out;
    int32_t* blk_status;
};

#define KNZ_ANS0_DEC_CHUNKS 8

struct KnzAns0DecShared {
    uint8_t s_f2s[KNZ_ANS0_DEC_CHUNKS][4096];
    uint32_t s_sym[KNZ_ANS0_DEC_CHUNKS][256];      // freq | cumFreq << 16
    uint16_t s_pay[KNZ_ANS0_DEC_CHUNKS][256];       // renormalisation words of each chunk, ring
    uint8_t s_ob[KNZ_ANS0_DEC_CHUNKS][256];         // decoded bytes of each chunk, 64 steps at a time
    uint16_t s_freqs[KNZ_ANS0_DEC_CHUNKS][256];
    uint8_t s_alphas[KNZ_ANS0_DEC_CHUNKS][256];
    uint32_t s_state[KNZ_ANS0_DEC_CHUNKS][4];
    uint64_t s_paybit[KNZ_ANS0_DEC_CHUNKS];
    uint64_t s_cbit[KNZ_ANS0_DEC_CHUNKS];           // first bit of each chunk (filled by the caller)
    int s_mode[KNZ_ANS0_DEC_CHUNKS];                // 0 absent, 1 raw, 2 single symbol, 3 rANS, -1 error
};

// chunk slots slotBase .. slotBase + nvalid - 1 (<= 8, all of them below a.nslots) by one wave
__device__ __forceinline__ void knz_ans0_decode_body(const Ans0DecArgs& a, const uint32_t slotBase, const uint32_t nvalid, KnzAns0DecShared& sh) {
    uint8_t (&s_f2s)[KNZ_ANS0_DEC_CHUNKS][4096] = sh.s_f2s;
    uint32_t (&s_sym)[KNZ_ANS0_DEC_CHUNKS][256] = sh.s_sym;
    uint16_t (&s_pay)[KNZ_ANS0_DEC_CHUNKS][256] = sh.s_pay;
    uint8_t (&s_ob)[KNZ_ANS0_DEC_CHUNKS][256] = sh.s_ob;
    uint16_t (&s_freqs)[KNZ_ANS0_DEC_CHUNKS][256] = sh.s_freqs;
    uint8_t (&s_alphas)[KNZ_ANS0_DEC_CHUNKS][256] = sh.s_alphas;
    uint32_t (&s_state)[KNZ_ANS0_DEC_CHUNKS][4] = sh.s_state;
    uint64_t (&s_paybit)[KNZ_ANS0_DEC_CHUNKS] = sh.s_paybit;
    const uint64_t (&s_cbit)[KNZ_ANS0_DEC_CHUNKS] = sh.s_cbit;
    int (&s_mode)[KNZ_ANS0_DEC_CHUNKS] = sh.s_mode;

    const int lane = threadIdx.x;
    const uint32_t cpb = a.chunks_per_block;
    const uint64_t limit = a.nbytes << 3;

    // ---- the 8 chunk headers, one lane each (the parse is a serial bit walk: 8 in flight instead of 8 in a row) ------------
    for (int i = lane; i < KNZ_ANS0_DEC_CHUNKS * 256; i += 64) (&s_freqs[0][0])[i] = 0;
    wave_sync();
    if (lane < KNZ_ANS0_DEC_CHUNKS) {
        const int cg = lane;
        uint16_t* s_freq = s_freqs[cg];
        uint8_t* s_alpha = s_alphas[cg];
        const uint32_t slotId = slotBase + cg;
        int mode = 0;
        if ((uint32_t)cg < nvalid) {
            const uint32_t b = slotId / cpb, k = slotId % cpb;
            const uint32_t preLen = a.blk_pre_len[b];
            if (a.blk_status[b] == 0 && (uint64_t)k * KNZ_ANS_CHUNK < preLen) mode = ((a.blk_mode[b] & 0x80) || preLen <= 32) ? 1 : 3;
        }
        if (mode == 3) {
            {                                                      // decodeHeader :605-710
                KnzStreamReader r;
                r.init(a.stream, a.nbytes, s_cbit[cg]);
                const uint32_t lr = 8 + r.read(3);
                int count = 0;
                int m = 3;
                if (lr != 12) m = -1;                             // the encoder only produces log range 12 for order 0
                else {
                    if (r.read(1) == 0) {
                        if (r.read(1) == 1) count = 0;
                        else { count = 256; for (int i = 0; i < 256; i++) s_alpha[i] = (uint8_t)i; }
                    } else {
                        const uint32_t lastMask = r.read(5);
                        for (uint32_t mm = 0; mm <= lastMask; mm++) {
                            const uint32_t mask = r.read(8);
                            for (int j = 0; j < 8; j++) if ((mask >> j) & 1) s_alpha[count++] = (uint8_t)(8 * mm + j);
                        }
                    }
                    if (count == 0) m = -1;
                    else {
                        const int chk = count < 64 ? 6 : 8;
                        int sum = 0;
                        for (int i = 1; i < count && m == 3; i += chk) {
                            const uint32_t logMax = r.read(4);
                            if ((1u << logMax) > 4096u) { m = -1; break; }
                            const int endj = min(i + chk, count);
                            for (int j = i; j < endj; j++) {
                                int fr = 1;
                                if (logMax > 0) { fr = 1 + (int)r.read(logMax); if (fr <= 0 || fr >= 4096) { m = -1; break; } }
                                s_freq[s_alpha[j]] = (uint16_t)fr;
                                sum += fr;
                            }
                        }
                        if (m == 3) {
                            if (4096 <= sum) m = -1;
                            else {
                                s_freq[s_alpha[0]] = (uint16_t)(4096 - sum);
                                if (count == 1) m = 2;
                                else {
                                    const uint32_t sz = knz_read_varint(r);
                                    if (sz >= (1u << 27)) m = -1;
                                    for (int c = 0; c < 4; c++) s_state[cg][c] = r.read(32);
                                    s_paybit[cg] = r.tell();
                                    if (r.tell() + 8ull * sz > limit + 7) m = -1;
                                }
                            }
                        }
                    }
                }
                mode = m;
            }
        }
        s_mode[cg] = mode;
    }
    wave_sync();

    for (int cg = 0; cg < KNZ_ANS0_DEC_CHUNKS; cg++) {
        const uint16_t* s_freq = s_freqs[cg];
        const uint8_t* s_alpha = s_alphas[cg];
        const uint32_t slotId = slotBase + cg;
        const int mode = s_mode[cg];
        uint32_t n = 0, b = 0, k = 0;
        if (mode != 0) {
            b = slotId / cpb; k = slotId % cpb;
            n = min((uint32_t)KNZ_ANS_CHUNK, a.blk_pre_len[b] - k * KNZ_ANS_CHUNK);
        }
        {
            if (mode == 3) {
                // cumulated frequencies (symbol order) with a wave scan over 4 symbols per lane, then the tables
                uint32_t f4[4], tot = 0;
                for (int j = 0; j < 4; j++) { f4[j] = s_freq[lane * 4 + j]; tot += f4[j]; }
                uint32_t cum = wave_scan_incl(tot) - tot;
                for (int j = 0; j < 4; j++) {
                    const uint32_t fr = f4[j];
                    const uint32_t fclamp = fr < 4095u ? fr : 4095u;        // decSymbol.reset :972-977
                    s_sym[cg][lane * 4 + j] = fclamp | (cum << 16);
                    for (uint32_t q = 0; q < fr; q++) s_f2s[cg][cum + q] = (uint8_t)(lane * 4 + j);
                    cum += fr;
                }
            } else if (mode == 2) {
                if (lane == 0) s_state[cg][0] = s_alpha[0];
            }
            wave_sync();
        }
        // raw / single-symbol chunks are finished here by the whole wave
        if (mode == 1 || mode == 2) {
            uint8_t* dst = a.out + a.blk_out_off[b] + (size_t)k * KNZ_ANS_CHUNK;
            if (mode == 2) { const uint8_t v = (uint8_t)s_state[cg][0]; for (uint32_t i = lane; i < n; i += 64) dst[i] = v; }
            else {
                const uint64_t cbit = s_cbit[cg];
                for (uint32_t i = lane * 4; i < n; i += 256) {
                    const uint32_t w = knz_fetch32(a.stream, (int64_t)(cbit + 8ull * i), (int64_t)limit);
                    for (uint32_t j = 0; j < 4 && i + j < n; j++) dst[i + j] = (uint8_t)(w >> (24 - 8 * j));
                }
            }
        } else if (mode == -1 && lane == 0) {
            a.blk_status[b] = KNZ_ERR_PROCESS_BLOCK;
        }
        wave_sync();
    }

    // ---- the 8 x 4 states ------------------------------------------------------------------------------------------------
    const int g = lane >> 2, c = lane & 3;
    const bool lanes32 = lane < 4 * KNZ_ANS0_DEC_CHUNKS;
    const uint32_t slotId = slotBase + (lanes32 ? g : 0);
    bool live = lanes32 && (uint32_t)g < nvalid && s_mode[lanes32 ? g : 0] == 3;
    uint32_t n = 0;
    uint8_t* dst = a.out;
    uint64_t paybit = 0;
    uint32_t st = 0;
    if (live) {
        const uint32_t b = slotId / cpb, k = slotId % cpb;
        const uint32_t preLen = a.blk_pre_len[b];
        n = min((uint32_t)KNZ_ANS_CHUNK, preLen - k * KNZ_ANS_CHUNK);
        dst = a.out + a.blk_out_off[b] + (size_t)k * KNZ_ANS_CHUNK;
        paybit = s_paybit[g];
        st = s_state[g][c];
    }
    const uint32_t end4 = n & ~3u;
    const uint32_t T = live ? (end4 >> 2) : 0;
    uint32_t maxT = T;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = wave_shfl(maxT, lane ^ d); maxT = o > maxT ? o : maxT; }
    uint32_t cnt = 0;
    const int gshift = (lane >> 2) << 2;
    // Neither the renormalisation words nor the decoded bytes touch global memory inside the dependent loop: the payload of
    // each chunk is staged 128 words at a time into an LDS ring (s_pay), the output leaves through a 256-byte LDS row per
    // chunk every 64 steps (a global access in the loop would put its latency, and the in-order vmcnt, on every step).
    uint32_t payHi = 0;                                                  // words [payHi - 256, payHi) of my chunk are staged
    for (uint32_t t = 0; t < maxT; t++) {
        if (wave_ballot(live && cnt + 4 > payHi) != 0) {                 // some chunk of the wave is about to run dry
            wave_sync();
            for (int cg = 0; cg < KNZ_ANS0_DEC_CHUNKS; cg++) {
                const uint32_t hiC = wave_readlane(payHi, 4 * cg), cntC = wave_readlane(cnt, 4 * cg);
                if (cntC + 4 <= hiC || wave_readlane((uint32_t)live, 4 * cg) == 0) continue;
                const uint64_t pb = wave_shfl64(paybit, 4 * cg);
                for (uint32_t j = lane; j < 128; j += 64) {
                    const uint32_t wi = hiC + j;
                    s_pay[cg][wi & 255] = (uint16_t)(knz_fetch32(a.stream, (int64_t)(pb + 16ull * wi), (int64_t)limit) >> 16);
                }
            }
            if (live && cnt + 4 > payHi) payHi += 128;
            wave_sync();
        }
        const bool act = t < T;
        uint32_t need = 0;
        if (act) {
            const uint32_t slot = st & 4095u;
            const uint32_t sym = s_f2s[g][slot];
            s_ob[g][(4 * t + (3 - c)) & 255] = (uint8_t)sym;              // block[i]=cur3 .. block[i+3]=cur0 (:908-919)
            const uint32_t e = s_sym[g][sym];
            st = (e & 0xFFFFu) * (st >> 12) + slot - (e >> 16);        // decodeSymbol :846-858
            need = st < (1u << 15) ? 1u : 0u;
        }
        const uint64_t bal = wave_ballot(need != 0);
        const uint32_t gb = (uint32_t)(bal >> gshift) & 0xFu;
        if (need) {
            // refill order inside one iteration: st3, st2, st1, st0
            const uint32_t r = cnt + (uint32_t)__popc(gb >> (c + 1));
            st = (st << 16) | s_pay[g][r & 255];
        }
        cnt += (uint32_t)__popc(gb);
        if ((t & 63) == 63 || t + 1 == maxT) {                           // 256 decoded bytes per chunk (or the rest)
            wave_sync();
            const uint32_t base = (t & ~63u) * 4;
            for (int cg = 0; cg < KNZ_ANS0_DEC_CHUNKS; cg++) {
                const uint32_t Tc = wave_readlane(T, 4 * cg);
                if (4 * Tc <= base) continue;
                const uint32_t m = min(256u, 4 * Tc - base);
                uint8_t* dc = (uint8_t*)wave_shfl64((uint64_t)dst, 4 * cg);
                for (uint32_t i = lane; i < m; i += 64) dc[base + i] = s_ob[cg][i];
            }
            wave_sync();
        }
    }
    if (live && c == 0) {
        for (uint32_t i = end4; i < n; i++)
            dst[i] = (uint8_t)(knz_fetch32(a.stream, (int64_t)(paybit + 16ull * cnt + 8ull * (i - end4)), (int64_t)limit) >> 24);
    }
}

__global__ __launch_bounds__(64) void knz_ans0_decode_kernel(Ans0DecArgs a) {
    __shared__ KnzAns0DecShared sh;
    const uint32_t slotBase = blockIdx.x * KNZ_ANS0_DEC_CHUNKS;
    const uint32_t nvalid = slotBase < a.nslots ? min((uint32_t)KNZ_ANS0_DEC_CHUNKS, a.nslots - slotBase) : 0u;
    if (threadIdx.x < KNZ_ANS0_DEC_CHUNKS) sh.s_cbit[threadIdx.x] = threadIdx.x < nvalid ? a.chunk_bit[slotBase + threadIdx.x] : 0;
    wave_sync();
    knz_ans0_decode_body(a, slotBase, nvalid, sh);
}

// Walk and decode in ONE launch, as for Huffman (knz_huf_walk_decode_kernel): workgroups [0, nblocks) walk (one wave each: the
// rANS chunk headers are walked without the ring), workgroup nblocks + kg * nblocks + b decodes chunks 8 kg .. 8 kg + 7 of block
// b; lanes 0..7 poll the positions of its chunks. The decode (4.7 ms on config 3's entropy half) is longer than the walk (2.4 ms)
// here, so it is the walk that disappears.
__global__ __launch_bounds__(64) void knz_ans0_walk_decode_kernel(WalkBlocksArgs wa, Ans0DecArgs da) {
    __shared__ union KnzAns0WalkDecodeShared { KnzAns0DecShared d; KnzWalkShared w; } sh;
    const uint32_t nblocks = wa.nblocks;
    const int lane = threadIdx.x;
    if (blockIdx.x < nblocks) {
        if (lane < 4) sh.w.s_sync[lane] = 0;
        __syncthreads();
        knz_walk_block_body<true>(wa, blockIdx.x, sh.w);
        return;
    }
    const uint32_t cpb = da.chunks_per_block;
    const uint32_t id = blockIdx.x - nblocks;
    const uint32_t kg = id / nblocks, b = id % nblocks;
    const uint32_t k0 = kg * KNZ_ANS0_DEC_CHUNKS;
    if (k0 >= cpb) return;
    const uint32_t nvalid = min((uint32_t)KNZ_ANS0_DEC_CHUNKS, cpb - k0);
    const uint32_t slotBase = b * cpb + k0;
    if (lane < KNZ_ANS0_DEC_CHUNKS) {
        uint64_t v = 0;
        const bool present = (uint32_t)lane < nvalid && da.blk_status[b] == 0 && (uint64_t)(k0 + lane) * KNZ_ANS_CHUNK < da.blk_pre_len[b];
        if (present) {
            v = knz_poll64(&da.chunk_bit[slotBase + lane]);
            for (uint32_t spins = 0; v == KNZ_CHUNK_NOT_READY && spins < (1u << 21); spins++) { wg_spin_pause(); v = knz_poll64(&da.chunk_bit[slotBase + lane]); }
            if (v == KNZ_CHUNK_NOT_READY) da.blk_status[b] = KNZ_ERR_PROCESS_BLOCK;     // (never seen: the walkers are resident first)
        }
        sh.d.s_cbit[lane] = v;
    }
    wave_sync();
    // a chunk whose walk failed (ERR mark) or never came is left out: its block carries the error
    uint32_t ok = nvalid;
    for (uint32_t c = 0; c < nvalid; c++) if (sh.d.s_cbit[c] >= KNZ_CHUNK_ERR) { ok = c; break; }
    knz_ans0_decode_body(da, slotBase, ok, sh.d);
}
