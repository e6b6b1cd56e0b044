// Static rANS order 1 of kanzi bitstream v6 on gfx950: 4 MiB chunks (min(16384<<8, 1<<27), ANSRangeCodec.go:98-100,153-155),
// log range 11, 256 context tables, the chunk split into 4 quarters coded BACKWARDS in lock-step by the 4 states, each
// quarter's first symbol coded in context 0 (ANSRangeCodec.go:353-388, rebuildStatistics :414-423).
//
//   knz_ans1_hist_kernel    4 workgroups per chunk, each owns 64 contexts in LDS and streams the chunk (L2/MALL resident)
//   knz_ans1_stats_kernel   one wave per (chunk, context): NormalizeFrequencies to 2048, symbol parameters, header bits
//   knz_ans1_merge_kernel   one workgroup per chunk: bit-granular concatenation of the 256 context headers (unit 0)
//   knz_ans1_expand_kernel  parallel: resolves the (context, symbol) -> parameters look-up of every coding step into a stream
//   knz_ans1_encode_kernel  one wave per chunk, lanes 0..3 = the 4 states; only the state-dependent arithmetic is serial;
//                           renormalisation words are placed with the ballot/popcount scheme of ans0.hip
// Decode: knz_ans1_dec_tables_kernel (header parse, slot table {symbol,freq,slot-cum} 2 MiB per chunk) and
// knz_ans1_decode_kernel (one lane per state, forward).
#include "bits.h"

#define KNZ_ANS1_CHUNK (4u << 20)
#define KNZ_ANS1_LR 11
#define KNZ_ANS1_SCALE 2048
#define KNZ_ANS1_LDS_MAX_CHUNKS 1280u                     // batches up to this many chunks decode from LDS (knz_host_api.inc)
#define KNZ_ANS1_CTXHDR_BYTES 448                     // 3454 bits worst case per context header
#define KNZ_ANS1_U0_CAP (256 * KNZ_ANS1_CTXHDR_BYTES + 64)
#define KNZ_ANS1_U1_OFF KNZ_ANS1_U0_CAP
#define KNZ_ANS1_PAY_OFF (KNZ_ANS1_U0_CAP + 64)
#define KNZ_ANS1_PAY_CAP (((KNZ_ANS1_CHUNK / 8) * 11) + 64)   // a symbol costs at most log2(2048) = 11 bits
#define KNZ_ANS1_SLOT (KNZ_ANS1_PAY_OFF + KNZ_ANS1_PAY_CAP + 64)
#define KNZ_ANS1_CUM_STRIDE 257                       // cum[ctx][0..256]
#define KNZ_ANS1_PAYRING 8192                         // 16-bit payload words staged in LDS by the LDS decoder
#define KNZ_ANS1_RING 2048                            // 16-bit words per chunk ring (power of two)

struct Ans1Args {
    const uint64_t* blk_off;       // absolute device address of each block's post-transform bytes
    const uint32_t* blk_len;
    uint32_t chunks_per_block;     // 4 MiB chunk slots per block
    uint32_t nslots;
    uint8_t* scratch;              // [nslots * KNZ_ANS1_SLOT]
    uint32_t* unit_bits;           // [nslots * 5]
    uint32_t* unit_src;            // [nslots * 5]
    uint32_t* freqs;               // [nslots * 256 * 256] order-1 counts
    uint2* tab;                    // [nslots * 65536] encoder symbol table
    uint8_t* ctx_hdr;              // [nslots * 256 * KNZ_ANS1_CTXHDR_BYTES] per-context header bits (BE words)
    uint32_t* ctx_bits;            // [nslots * 256]
    int32_t* blk_status;
};

__device__ __forceinline__ bool knz_ans1_chunk(const Ans1Args& a, uint32_t slotId, uint32_t& b, uint32_t& n, const uint8_t*& src) {
    b = slotId / a.chunks_per_block;
    const uint32_t k = slotId % a.chunks_per_block;
    const uint32_t postLen = a.blk_len[b];
    if ((uint64_t)k * KNZ_ANS1_CHUNK >= postLen || postLen <= 32) return false;      // <= 32 bytes: raw (ANSRangeCodec.go:279-282)
    n = min(KNZ_ANS1_CHUNK, postLen - k * KNZ_ANS1_CHUNK);
    src = (const uint8_t*)a.blk_off[b] + (size_t)k * KNZ_ANS1_CHUNK;
    return true;
}

// order-1 histogram with the quarter rule (Global.go:252-299 called per quarter, ANSRangeCodec.go:414-423): 4 x 32 workgroups per
// chunk: workgroup (grp, slice) counts the contexts 64*grp.. of one slice of the chunk in 64 KiB of LDS and adds its non-zero
// counters to the (zeroed) chunk table
#define KNZ_ANS1_HIST_WGS 4
#define KNZ_ANS1_HIST_SLICES 32
#define KNZ_ANS1_HIST_CTX (256 / KNZ_ANS1_HIST_WGS)
// (Round 4: a thread takes 16 positions per trip, one 16-byte read + the byte in front, and a chunk is cut into 32 slices: the first form read
// two single bytes per trip with two waves per SIMD, 2048 dependent trips of ~1 us: 1.9 ms for 55 MB, nothing to do with the counters.)
__global__ __launch_bounds__(256) void knz_ans1_hist_kernel(Ans1Args a) {
    __shared__ uint32_t s_h[KNZ_ANS1_HIST_CTX][256];
    const int tid = threadIdx.x;
    const uint32_t slotId = blockIdx.x / (KNZ_ANS1_HIST_WGS * KNZ_ANS1_HIST_SLICES);
    const uint32_t grp = (blockIdx.x / KNZ_ANS1_HIST_SLICES) % KNZ_ANS1_HIST_WGS, slice = blockIdx.x % KNZ_ANS1_HIST_SLICES;
    uint32_t b, n; const uint8_t* src;
    if (!knz_ans1_chunk(a, slotId, b, n, src)) return;
    const uint32_t quarter = n >> 2;
    const uint32_t counted = quarter == 0 ? n : 4 * quarter;      // the (n & 3) tail is stored raw
    const uint32_t per = max(4096u, (((counted + KNZ_ANS1_HIST_SLICES - 1) / KNZ_ANS1_HIST_SLICES) + 15u) & ~15u);   // (a slice is at least one trip of the workgroup)
    const uint32_t lo = min(counted, slice * per), hi = min(counted, lo + per);
    if (lo >= hi) return;
    for (int i = tid; i < KNZ_ANS1_HIST_CTX * 256; i += 256) (&s_h[0][0])[i] = 0;
    __syncthreads();
    for (uint32_t p0 = lo + 16u * tid; p0 < hi; p0 += 16u * 256u) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        uint32_t prev = p0 ? (uint32_t)src[p0 - 1] : 0u;
        if (p0 + 16u <= hi) { const uint64_t a0 = knz_vle64(src + p0), a1 = knz_vle64(src + p0 + 8); w[0] = (uint32_t)a0; w[1] = (uint32_t)(a0 >> 32); w[2] = (uint32_t)a1; w[3] = (uint32_t)(a1 >> 32); }
        else for (uint32_t j = 0; j < 16u; j++) if (p0 + j < hi) w[j >> 2] |= (uint32_t)src[p0 + j] << (8u * (j & 3u));
        const uint32_t cnt = min(16u, hi - p0);
        // a quarter's first symbol is counted in context 0: which of the 16 positions is one (at most one unless quarters are shorter than 16)
        uint32_t firsts = p0 == 0 ? 1u : 0u;
        if (quarter != 0) for (uint32_t k = 1; k < 4u; k++) { const uint32_t d = k * quarter - p0; if (d < 16u) firsts |= 1u << d; }
#pragma unroll
        for (uint32_t j = 0; j < 16u; j++) {
            const uint32_t sym = (w[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
            const uint32_t ctx = ((firsts >> j) & 1u) ? 0u : prev;
            if (j < cnt && ctx / KNZ_ANS1_HIST_CTX == grp) atomicAdd(&s_h[ctx % KNZ_ANS1_HIST_CTX][sym], 1u);
            prev = sym;
        }
    }
    __syncthreads();
    uint32_t* out = a.freqs + ((size_t)slotId * 256 + grp * KNZ_ANS1_HIST_CTX) * 256;
    for (int i = tid; i < KNZ_ANS1_HIST_CTX * 256; i += 256) { const uint32_t v = (&s_h[0][0])[i]; if (v) atomicAdd(&out[i], v); }
}

// one wave per (chunk, context)
__global__ __launch_bounds__(64) void knz_ans1_stats_kernel(Ans1Args a) {
    __shared__ int s_f[256];
    __shared__ int s_alpha[256];
    __shared__ uint32_t s_hdr[KNZ_ANS1_CTXHDR_BYTES / 4];
    __shared__ int s_asize, s_panic;
    __shared__ uint32_t s_bits;
    const int lane = threadIdx.x;
    const uint32_t slotId = blockIdx.x >> 8, ctx = blockIdx.x & 255;
    uint32_t b, n; const uint8_t* src;
    if (!knz_ans1_chunk(a, slotId, b, n, src)) return;
    const uint32_t* fr = a.freqs + ((size_t)slotId * 256 + ctx) * 256;
    uint32_t tot = 0;
    for (int j = 0; j < 4; j++) { const uint32_t v = fr[lane * 4 + j]; s_f[lane * 4 + j] = (int)v; tot += v; }
    tot = wave_reduce_add(tot);
    for (int i = lane; i < KNZ_ANS1_CTXHDR_BYTES / 4; i += 64) s_hdr[i] = 0;
    wave_sync();
    if (lane == 0) {
        int panic = 0;
        s_asize = knz_normalize_freqs(s_f, 256, s_alpha, (int)tot, KNZ_ANS1_SCALE, &panic);   // updateFrequencies :185
        s_panic = panic;
    }
    wave_sync();
    const int asize = s_asize;
    // encSymbol.reset for the 4 symbols of this lane (:446-468)
    uint32_t f4[4], sum4 = 0;
    for (int j = 0; j < 4; j++) { f4[j] = (uint32_t)s_f[lane * 4 + j]; sum4 += f4[j]; }
    uint32_t cum = wave_scan_incl(sum4) - sum4;
    uint2* tab = a.tab + ((size_t)slotId * 256 + ctx) * 256;
    for (int j = 0; j < 4; j++) {
        const uint32_t f = f4[j];
        uint2 e; e.x = 0; e.y = 0;
        if (f != 0 && asize > 0) {
            const uint32_t frq = f < (KNZ_ANS1_SCALE - 1) ? f : (KNZ_ANS1_SCALE - 1);
            uint32_t bias, invFreq, sh;
            if (frq < 2) { invFreq = 0xFFFFFFFFu; sh = 0; bias = cum + (KNZ_ANS1_SCALE - 1); }
            else {
                uint32_t shift = 0;
                while (frq > (1u << shift)) shift++;
                invFreq = (uint32_t)(((((uint64_t)1) << (shift + 31)) + (uint64_t)(frq - 1)) / (uint64_t)frq);
                sh = shift - 1;
                bias = cum;
            }
            e.x = frq | (bias << 12) | (sh << 25);
            e.y = invFreq;
        }
        tab[lane * 4 + j] = e;
        cum += f;
    }
    if (lane == 0) {   // encodeHeader :216-270 for this context
        KnzBitWriter bw;
        bw.init(s_hdr);
        if (asize == 256) { bw.put(0, 1); bw.put(0, 1); }
        else if (asize == 0) { bw.put(0, 1); bw.put(1, 1); }
        else {
            bw.put(1, 1);
            const int lastMask = s_alpha[asize - 1] >> 3;
            bw.put((uint32_t)lastMask, 5);
            for (int m = 0; m <= lastMask; m++) {
                uint32_t mask = 0;
                for (int bit = 0; bit < 8; bit++) mask |= (s_f[8 * m + bit] != 0 ? 1u : 0u) << bit;
                bw.put(mask, 8);
            }
        }
        if (asize > 1) {
            const int chk = asize < 64 ? 6 : 8;
            const uint32_t llr = 4;                    // smallest llr with 1<<llr > 11
            for (int i = 1; i < asize; i += chk) {
                int mx = s_f[s_alpha[i]] - 1;
                const int endj = min(i + chk, asize);
                for (int j = i + 1; j < endj; j++) { const int v = s_f[s_alpha[j]] - 1; if (v > mx) mx = v; }
                uint32_t logMax = 0;
                while ((1 << logMax) <= mx) logMax++;
                bw.put(logMax, llr);
                if (logMax == 0) continue;
                for (int j = i; j < endj; j++) bw.put((uint32_t)(s_f[s_alpha[j]] - 1), logMax);
            }
        }
        s_bits = bw.pos;
    }
    wave_sync();
    if (s_panic && lane == 0) a.blk_status[b] = KNZ_ERR_PROCESS_BLOCK;
    uint32_t* g = (uint32_t*)(a.ctx_hdr + ((size_t)slotId * 256 + ctx) * KNZ_ANS1_CTXHDR_BYTES);
    const uint32_t hb = s_bits;
    for (uint32_t i = lane; i < ((hb + 31) >> 5); i += 64) g[i] = knz_bswap32(s_hdr[i]);
    if (lane == 0) a.ctx_bits[(size_t)slotId * 256 + ctx] = hb;
}

// unit 0 = (lr-8):3 then the 256 context headers back to back
__global__ __launch_bounds__(256) void knz_ans1_merge_kernel(Ans1Args a) {
    __shared__ uint32_t s_start[257];
    __shared__ uint32_t s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t slotId = blockIdx.x;
    uint32_t* ubits = a.unit_bits + (size_t)slotId * KNZ_UNITS_PER_CHUNK;
    uint32_t* usrc = a.unit_src + (size_t)slotId * KNZ_UNITS_PER_CHUNK;
    if (tid < KNZ_UNITS_PER_CHUNK) { ubits[tid] = 0; usrc[tid] = tid == 1 ? KNZ_ANS1_U1_OFF : (tid == 2 ? KNZ_ANS1_PAY_OFF : 0u); }
    uint32_t b, n; const uint8_t* src;
    uint8_t* slot = a.scratch + (size_t)slotId * KNZ_ANS1_SLOT;
    const uint32_t bb = slotId / a.chunks_per_block, kk = slotId % a.chunks_per_block;
    if (!knz_ans1_chunk(a, slotId, b, n, src)) {
        // whole input <= 32 bytes: raw bytes as unit 0 of the block's first slot
        const uint32_t postLen = a.blk_len[bb];
        if (kk == 0 && postLen <= 32 && postLen > 0) {
            const uint8_t* s = (const uint8_t*)a.blk_off[bb];
            if (tid < (int)postLen) slot[tid] = s[tid];
            if (tid == 0) ubits[0] = 8 * postLen;
        }
        return;
    }
    const uint32_t myBits = a.ctx_bits[(size_t)slotId * 256 + tid];
    const uint32_t incl = wave_scan_incl(myBits);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t off = 3 + incl - myBits;
    for (int w = 0; w < wave; w++) off += s_wave[w];
    s_start[tid] = off;
    if (tid == 255) s_start[256] = off + myBits;
    __syncthreads();
    const uint32_t total = s_start[256];
    uint32_t* out = (uint32_t*)slot;
    const uint8_t* hdrs = a.ctx_hdr + (size_t)slotId * 256 * KNZ_ANS1_CTXHDR_BYTES;
    for (uint32_t w = tid; w < ((total + 31) >> 5); w += 256) {
        const uint32_t wbit = w << 5;
        uint32_t v = (w == 0) ? ((uint32_t)(KNZ_ANS1_LR - 8) << 29) : 0u;
        // first context whose range ends after wbit
        int lo = 0, hi = 256;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_start[mid + 1] > wbit) hi = mid; else lo = mid + 1; }
        for (int c = lo; c < 256 && s_start[c] < wbit + 32; c++) {
            const uint32_t nb = s_start[c + 1] - s_start[c];
            if (nb == 0) continue;
            v |= knz_fetch32_unit(hdrs + (size_t)c * KNZ_ANS1_CTXHDR_BYTES, 0, (int64_t)wbit - (int64_t)s_start[c], nb);
        }
        out[w] = knz_bswap32(v);
    }
    if (tid == 0) ubits[0] = total;
}

// Expanded per-step coder parameters: the sequence of (context, symbol) pairs a state walks does not depend on the state
// value, so a fully parallel pass resolves the two dependent look-ups (byte -> table entry) for every step up front and
// leaves them as a sequential 16-byte stream: entry (t, c) at ent[slot][4 * t + c] =
//   { invFreq, xMax = freq << 20, bias, (2048 - freq) << 8 | invShift }.
// Steps past the end are padding {0, 0xFFFFFFFF, 0, 0}: never renormalise, state unchanged.
#define KNZ_ANS1_GROUP 16                              // steps per register buffer of the serial kernel
#define KNZ_ANS1_ENT_STEPS ((KNZ_ANS1_CHUNK >> 2) + 128)
#define KNZ_ANS1_ENT_STRIDE ((size_t)KNZ_ANS1_ENT_STEPS * 4)    // uint4 entries per chunk slot

// number of coding steps of a chunk (0: nothing coded), and whether the Go code would panic on it
__device__ __forceinline__ uint32_t knz_ans1_steps(const Ans1Args& a, uint32_t slotId, uint32_t& b, uint32_t& n, const uint8_t*& src, bool& bad) {
    bad = false; b = 0; n = 0; src = nullptr;
    if (!knz_ans1_chunk(a, slotId, b, n, src)) { n = 0; return 0; }
    if (a.blk_status[b] != 0) { n = 0; return 0; }
    const uint32_t q = n >> 2;
    if (n > 1 && q == 0) { bad = true; n = 0; return 0; }   // 2..3 byte chunk: Go indexes block[-1] and panics
    return n > 1 ? q : 0;                                    // encodeChunk: `else if len(block) > 1` (:353)
}

__device__ __forceinline__ uint32_t knz_ans1_padded_steps(uint32_t steps) {
    return ((steps + 5 * KNZ_ANS1_GROUP - 1) / (3 * KNZ_ANS1_GROUP)) * (3 * KNZ_ANS1_GROUP) + KNZ_ANS1_GROUP;
}

__global__ __launch_bounds__(256) void knz_ans1_expand_kernel(Ans1Args a, uint4* ent) {
    const uint32_t slotId = blockIdx.x;
    uint32_t b, n; const uint8_t* src; bool bad;
    const uint32_t steps = knz_ans1_steps(a, slotId, b, n, src, bad);
    if (steps == 0) return;
    const uint32_t q = steps;
    const uint32_t total = knz_ans1_padded_steps(steps) * 4;
    const uint2* __restrict__ tab = a.tab + (size_t)slotId * 65536;
    uint4* out = ent + (size_t)slotId * KNZ_ANS1_ENT_STRIDE;
    // four entries per thread and trip, the reads of all four issued before the first look-up (one at a time a trip is three dependent
    // memory latencies, ~130 trips per thread: 0.86 ms for 55 MB where the 16 bytes written per symbol take 0.2)
    const uint32_t stride = gridDim.y * 256;
    for (uint32_t i0 = blockIdx.y * 256 + threadIdx.x; i0 < total; i0 += 4 * stride) {
        uint32_t key[4];
        uint32_t symv[4], ctxv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {                                        // (clamped addresses, no branch: the eight byte reads of a trip leave together)
            const uint32_t i = i0 + k * stride, t = min(i >> 2, steps - 1u), c = i & 3;
            // step t of state c codes symbol qbase[q-1-t] in context qbase[q-2-t] (context 0 for the quarter's first symbol)
            const uint8_t* qbase = src + (size_t)c * q;
            symv[k] = qbase[q - 1 - t];
            ctxv[k] = qbase[t + 1 < q ? q - 2 - t : 0u];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = i0 + k * stride, t = i >> 2;
            key[k] = (i < total && t < steps) ? ((t + 1 < q ? ctxv[k] << 8 : 0u) | symv[k]) : 0xFFFFFFFFu;
        }
        uint2 e[4];
#pragma unroll
        for (int k = 0; k < 4; k++) e[k] = tab[key[k] & 0xFFFFu];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t i = i0 + k * stride;
            if (i >= total) break;
            uint4 o; o.x = 0; o.y = 0xFFFFFFFFu; o.z = 0; o.w = 0;
            if (key[k] != 0xFFFFFFFFu) {
                const uint32_t freq = e[k].x & 0xFFFu, bias = (e[k].x >> 12) & 0x1FFFu, sh = (e[k].x >> 25) & 0xFu;
                o.x = e[k].y;
                o.y = freq << 20;                                   // xMax = ((ANS_TOP >> 11) << 16) * freq
                o.z = bias;
                o.w = (((uint32_t)KNZ_ANS1_SCALE - freq) << 16) | sh;     // (the factor in the high half: the multiply selects it as an operand half, no shift in the chain)
            }
            out[i] = o;
        }
    }
}

// One wave per chunk; lanes 0..3 carry the 4 states. Per step the dependent chain is: compare with xMax, place the 16-bit
// word (rank among the renormalising states via the ballot), the reciprocal multiply and the state update. The entries
// stream in through three register buffers of 16 steps (two groups in flight), renormalisation words are staged in an LDS
// ring and flushed by the whole wave: on gfx950 stores and loads share the in-order vmcnt counter, so a global store
// inside the dependent loop would stall every following entry load.
__global__ __launch_bounds__(64) void knz_ans1_encode_kernel(Ans1Args a, const uint4* ent) {
    __shared__ uint16_t s_w[3 * KNZ_ANS1_GROUP * 4];                    // candidate word of (step, state) of the current 48 steps
    const int lane = threadIdx.x;
    const uint32_t slotId = blockIdx.x;
    uint32_t b, n; const uint8_t* src; bool bad;
    const uint32_t steps = knz_ans1_steps(a, slotId, b, n, src, bad);
    const bool live = n != 0;
    uint8_t* slot = a.scratch + (size_t)slotId * KNZ_ANS1_SLOT;
    uint8_t* payEnd = slot + KNZ_ANS1_PAY_OFF + KNZ_ANS1_PAY_CAP;
    // lanes 4..63 mirror lanes 0..3 (same entries, same state, same LDS words): no divergence, no masked loads
    const uint4* __restrict__ my = ent + (size_t)slotId * KNZ_ANS1_ENT_STRIDE + (size_t)(lane & 3);
    uint32_t st = 1u << 15;
    uint32_t flushed = 0;
    uint4 buf0[KNZ_ANS1_GROUP], buf1[KNZ_ANS1_GROUP], buf2[KNZ_ANS1_GROUP];
    auto load_group = [&](uint32_t t0, uint4* e) {
#pragma unroll
        for (int j = 0; j < KNZ_ANS1_GROUP; j++) e[j] = my[(size_t)(t0 + j) * 4];
    };
    // Per step only the state arithmetic is on the dependent chain. Whether a state renormalises goes, as 4 bits per step,
    // into a wave-uniform 64-bit mask per group (scalar unit); the low 16 bits of every state go to a FIXED LDS slot of
    // (step, state). Which of those candidates are real words and where they land in the descending stream is sorted out
    // once per 48 steps by all 64 lanes (emit order = step, then state = bit order of the masks).
    auto run_group = [&](const uint4* e, uint16_t* wslot, uint64_t& mask) {
        mask = 0;
#pragma unroll
        for (int j = 0; j < KNZ_ANS1_GROUP; j++) {
            const bool x = st >= e[j].y;
            mask |= wave_ballot(x) & (0xFull << (4 * j));                    // lanes 4j..4j+3 speak for step j (every lane mirrors lanes 0..3): no shift
            wslot[4 * j + (lane & 3)] = (uint16_t)st;
            st = x ? (st >> 16) : st;
            // q = st / freq < 2^20 after the renormalisation (st < freq << 20): the product with 2048 - freq fits mul24
            const uint32_t qq = (uint32_t)(((uint64_t)st * e[j].x) >> 32) >> (e[j].w & 31u);
            st = st + e[j].z + knz_mul24(qq, e[j].w >> 16);
        }
    };
    auto flush = [&](uint64_t m0, uint64_t m1, uint64_t m2) {
        wave_sync_lds();
        const uint64_t below = ((uint64_t)1 << lane) - 1;
        const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1), c2 = (uint32_t)__popcll(m2);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint64_t m = r == 0 ? m0 : (r == 1 ? m1 : m2);
            const uint32_t before = r == 0 ? 0u : (r == 1 ? c0 : c0 + c1);
            if ((m >> lane) & 1) {
                const uint32_t pos = flushed + before + (uint32_t)__popcll(m & below);
                const uint16_t wv = s_w[64 * r + lane];
                uint8_t* p = payEnd - 2 * ((size_t)pos + 1);
                p[0] = (uint8_t)(wv >> 8);
                p[1] = (uint8_t)wv;
            }
        }
        flushed += c0 + c1 + c2;
        wave_sync_lds();
    };
    // The pipeline starts on neutral entries (two groups of no-ops) instead of a load prologue: every load of the kernel is
    // then issued inside the loop in program order, which is what lets the compiler wait with vmcnt(32+) instead of vmcnt(0).
#pragma unroll
    for (int j = 0; j < KNZ_ANS1_GROUP; j++) { buf0[j].x = 0; buf0[j].y = 0xFFFFFFFFu; buf0[j].z = 0; buf0[j].w = 0; buf1[j] = buf0[j]; }
    for (uint32_t t0 = 0; t0 < steps + 2 * KNZ_ANS1_GROUP && steps; t0 += 3 * KNZ_ANS1_GROUP) {
        uint64_t m0, m1, m2;
        load_group(t0, buf2); run_group(buf0, s_w, m0);
        load_group(t0 + KNZ_ANS1_GROUP, buf0); run_group(buf1, s_w + 4 * KNZ_ANS1_GROUP, m1);
        load_group(t0 + 2 * KNZ_ANS1_GROUP, buf1); run_group(buf2, s_w + 8 * KNZ_ANS1_GROUP, m2);
        flush(m0, m1, m2);
    }
    const uint32_t s1 = wave_shfl(st, 1), s2 = wave_shfl(st, 2), s3 = wave_shfl(st, 3);
    if (bad && lane == 0) a.blk_status[b] = KNZ_ERR_PROCESS_BLOCK;
    if (live && lane == 0) {
        const uint32_t end4 = n & ~3u;
        const uint32_t tail = n & 3;
        for (uint32_t i = 0; i < tail; i++) payEnd[i] = src[end4 + i];
        const uint32_t nbytes = 2 * flushed + tail;
        uint32_t w[8];
        for (int i = 0; i < 8; i++) w[i] = 0;
        KnzBitWriter bw;
        bw.init(w);
        knz_put_varint(bw, nbytes);
        bw.put(st, 32); bw.put(s1, 32); bw.put(s2, 32); bw.put(s3, 32);
        uint32_t* u1 = (uint32_t*)(slot + KNZ_ANS1_U1_OFF);
        for (int i = 0; i < 8; i++) u1[i] = knz_bswap32(w[i]);
        uint32_t* ubits = a.unit_bits + (size_t)slotId * KNZ_UNITS_PER_CHUNK;
        uint32_t* usrc = a.unit_src + (size_t)slotId * KNZ_UNITS_PER_CHUNK;
        ubits[1] = bw.pos;
        ubits[2] = 8 * nbytes;
        usrc[2] = (uint32_t)(KNZ_ANS1_PAY_OFF + KNZ_ANS1_PAY_CAP - 2 * flushed);
    }
}

#ifndef KNZ_HIP_EMU
// The same kernel with the dependent loop written by hand (device build; the emulator runs the C++ form above, which stays the A/B form on the
// device: KNZ_ANS1_ENC_PLAIN). A lone wave pays ~2.5 ns for every instruction it issues (profiles/r02_lone_wave_latencies.md); the compiler's loop
// has 11.5-12 per step: the renormalisation shift and its select are two instructions, every step folds its four ballot bits into a scalar mask
// (s_and + s_or / s_mov), and every step has its own s_waitcnt. Here a step is 6 instructions + a quarter of an LDS store + its entry load:
//   (no store)     the state BEFORE the renormalisation stays in its register (every step writes a new one) and leaves with three others in one
//                  16-byte LDS store per four steps, to the fixed slot of (state, step): its low half is the candidate word, and
//                  whether it is a real word is decided once per 48 steps by all 64 lanes (lane = 4 * step + state compares its slot with the
//                  xMax of its (step, state), which it loaded together with the group's entries): no mask bookkeeping on the chain;
//   v_cmp_ge_u32 / v_cndmask_b32_sdwa (src1_sel:WORD_1)   st = st >= xMax ? st >> 16 : st  in one select;
//   v_mul_hi_u32, v_lshrrev_b32, v_mul_u32_u24_sdwa, v_add3_u32   st += bias + (st / freq) * (2^11 - freq)  as before.
// Four steps per statement (an asm statement takes at most 30 operands); the compiler places the loads between the statements and one
// s_waitcnt in front of each (the loads of a group return in order).
#define KNZ_A1E_SDWA " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
// one step: state %[sA] -> %[sB]; %[sA] stays what it was (the state in front of the renormalisation: the candidate word and the flag test need it)
#define KNZ_A1E_STEP(J, A, B) \
    "v_cmp_ge_u32_e32 vcc, %[" A "], %[y" J "]\n\t" \
    "v_cndmask_b32_sdwa %[t], %[" A "], %[" A "], vcc" KNZ_A1E_SDWA \
    "v_mul_hi_u32 %[q], %[t], %[x" J "]\n\t" \
    "v_lshrrev_b32_e32 %[q], %[w" J "], %[q]\n\t" \
    "v_mul_u32_u24_sdwa %[q], %[q], %[w" J "]" KNZ_A1E_SDWA \
    "v_add3_u32 %[" B "], %[t], %[z" J "], %[q]\n\t"
// four steps; the four states in front of their renormalisations leave as ONE 16-byte LDS store (slot layout [group][state][step])
__device__ __forceinline__ void knz_a1e_run4(uint32_t& st, const uint4& e0, const uint4& e1, const uint4& e2, const uint4& e3, uint32_t* slot) {
    uint32_t q, t, s1, s2, s3, s4;
    asm volatile(KNZ_A1E_STEP("0", "s0", "s1") KNZ_A1E_STEP("1", "s1", "s2") KNZ_A1E_STEP("2", "s2", "s3") KNZ_A1E_STEP("3", "s3", "s4")
                 : [s1] "=&v"(s1), [s2] "=&v"(s2), [s3] "=&v"(s3), [s4] "=&v"(s4), [q] "=&v"(q), [t] "=&v"(t)
                 : [s0] "v"(st), [x0] "v"(e0.x), [y0] "v"(e0.y), [z0] "v"(e0.z), [w0] "v"(e0.w), [x1] "v"(e1.x), [y1] "v"(e1.y), [z1] "v"(e1.z), [w1] "v"(e1.w),
                   [x2] "v"(e2.x), [y2] "v"(e2.y), [z2] "v"(e2.z), [w2] "v"(e2.w), [x3] "v"(e3.x), [y3] "v"(e3.y), [z3] "v"(e3.z), [w3] "v"(e3.w)
                 : "vcc");
    uint4 sv; sv.x = st; sv.y = s1; sv.z = s2; sv.w = s3;
    *(uint4*)slot = sv;
    st = s4;
}
template <int R>
__device__ __forceinline__ void knz_a1e_run_group(uint32_t& st, const uint4* e, uint32_t* mine) {
    knz_a1e_run4(st, e[0], e[1], e[2], e[3], mine + R * 64);
    knz_a1e_run4(st, e[4], e[5], e[6], e[7], mine + R * 64 + 4);
    knz_a1e_run4(st, e[8], e[9], e[10], e[11], mine + R * 64 + 8);
    knz_a1e_run4(st, e[12], e[13], e[14], e[15], mine + R * 64 + 12);
}

__global__ __launch_bounds__(64) void knz_ans1_encode_asm_kernel(Ans1Args a, const uint4* ent) {
    __shared__ __attribute__((aligned(16))) uint32_t s_w[3 * KNZ_ANS1_GROUP * 4];   // state in front of the renormalisation of (group, state, step) of the current 48 steps
    const int lane = threadIdx.x;
    const uint32_t slotId = blockIdx.x;
    uint32_t b, n; const uint8_t* src; bool bad;
    const uint32_t steps = knz_ans1_steps(a, slotId, b, n, src, bad);
    const bool live = n != 0;
    uint8_t* slot = a.scratch + (size_t)slotId * KNZ_ANS1_SLOT;
    uint8_t* payEnd = slot + KNZ_ANS1_PAY_OFF + KNZ_ANS1_PAY_CAP;
    const uint4* __restrict__ base = ent + (size_t)slotId * KNZ_ANS1_ENT_STRIDE;
    const uint4* __restrict__ my = base + (size_t)(lane & 3);           // lanes 4..63 mirror lanes 0..3 (same entries, same state, same LDS words)
    uint32_t* mine = s_w + 16 * (lane & 3);                            // this lane's state: 16 steps of a group side by side
    uint32_t st = 1u << 15;
    uint32_t flushed = 0;
    uint4 buf0[KNZ_ANS1_GROUP], buf1[KNZ_ANS1_GROUP], buf2[KNZ_ANS1_GROUP];
    uint32_t xm0 = 0xFFFFFFFFu, xm1 = 0xFFFFFFFFu, xm2;               // xMax of (step = lane >> 2, state = lane & 3) of the group a buffer holds
    auto load_group = [&](uint32_t t0, uint4* e, uint32_t& xm) {
#pragma unroll
        for (int j = 0; j < KNZ_ANS1_GROUP; j++) e[j] = my[(size_t)(t0 + j) * 4];
        xm = base[(size_t)t0 * 4 + (uint32_t)lane].y;
    };
    auto flush = [&](uint32_t x0, uint32_t x1, uint32_t x2) {
        wave_sync_lds();
        const uint64_t below = ((uint64_t)1 << lane) - 1;
        const uint32_t fi = 16u * (uint32_t)(lane & 3) + (uint32_t)(lane >> 2);     // lane = 4 * step + state reads slot [state][step]
        const uint32_t v0 = s_w[fi], v1 = s_w[64 + fi], v2 = s_w[128 + fi];
        const uint64_t m0 = wave_ballot(v0 >= x0), m1 = wave_ballot(v1 >= x1), m2 = wave_ballot(v2 >= x2);
        const uint32_t c0 = (uint32_t)__popcll(m0), c1 = (uint32_t)__popcll(m1), c2 = (uint32_t)__popcll(m2);
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const uint64_t m = r == 0 ? m0 : (r == 1 ? m1 : m2);
            const uint32_t before = r == 0 ? 0u : (r == 1 ? c0 : c0 + c1);
            if ((m >> lane) & 1) {
                const uint32_t pos = flushed + before + (uint32_t)__popcll(m & below);
                const uint32_t wv = r == 0 ? v0 : (r == 1 ? v1 : v2);
                uint8_t* p = payEnd - 2 * ((size_t)pos + 1);
                p[0] = (uint8_t)(wv >> 8);
                p[1] = (uint8_t)wv;
            }
        }
        flushed += c0 + c1 + c2;
        wave_sync_lds();
    };
#pragma unroll
    for (int j = 0; j < KNZ_ANS1_GROUP; j++) { buf0[j].x = 0; buf0[j].y = 0xFFFFFFFFu; buf0[j].z = 0; buf0[j].w = 0; buf1[j] = buf0[j]; }
    for (uint32_t t0 = 0; t0 < steps + 2 * KNZ_ANS1_GROUP && steps; t0 += 3 * KNZ_ANS1_GROUP) {
        load_group(t0, buf2, xm2); knz_a1e_run_group<0>(st, buf0, mine);
        const uint32_t f0 = xm0;
        load_group(t0 + KNZ_ANS1_GROUP, buf0, xm0); knz_a1e_run_group<1>(st, buf1, mine);
        const uint32_t f1 = xm1;
        load_group(t0 + 2 * KNZ_ANS1_GROUP, buf1, xm1); knz_a1e_run_group<2>(st, buf2, mine);
        flush(f0, f1, xm2);
    }
    const uint32_t s1 = wave_shfl(st, 1), s2 = wave_shfl(st, 2), s3 = wave_shfl(st, 3);
    if (bad && lane == 0) a.blk_status[b] = KNZ_ERR_PROCESS_BLOCK;
    if (live && lane == 0) {
        const uint32_t end4 = n & ~3u;
        const uint32_t tail = n & 3;
        for (uint32_t i = 0; i < tail; i++) payEnd[i] = src[end4 + i];
        const uint32_t nbytes = 2 * flushed + tail;
        uint32_t w[8];
        for (int i = 0; i < 8; i++) w[i] = 0;
        KnzBitWriter bw;
        bw.init(w);
        knz_put_varint(bw, nbytes);
        bw.put(st, 32); bw.put(s1, 32); bw.put(s2, 32); bw.put(s3, 32);
        uint32_t* u1 = (uint32_t*)(slot + KNZ_ANS1_U1_OFF);
        for (int i = 0; i < 8; i++) u1[i] = knz_bswap32(w[i]);
        uint32_t* ubits = a.unit_bits + (size_t)slotId * KNZ_UNITS_PER_CHUNK;
        uint32_t* usrc = a.unit_src + (size_t)slotId * KNZ_UNITS_PER_CHUNK;
        ubits[1] = bw.pos;
        ubits[2] = 8 * nbytes;
        usrc[2] = (uint32_t)(KNZ_ANS1_PAY_OFF + KNZ_ANS1_PAY_CAP - 2 * flushed);
    }
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Decoder
struct Ans1DecArgs {
    const uint8_t* stream; uint64_t nbytes;
    const uint32_t* blk_pre_len;
    const uint8_t* blk_mode;
    const uint64_t* chunk_bit;
    const uint64_t* blk_out_off;   // absolute address of the block's output
    uint32_t chunks_per_block;
    uint32_t nslots;
    uint32_t* dtab;                // [nslots * 256 * 2048] sym | freq<<8 | (slot-cum)<<20
    uint32_t* info;                // [nslots * 8] {mode, st0..st3, lr}
    uint64_t* paybit;              // [nslots]
    int32_t* blk_status;
    uint32_t plain_loop;           // 0: the hand-written block; 1 (KNZ_ANS1_PLAIN): the compiler's schedule of the same loop; 2 (KNZ_ANS1_LOHI_LDS): round 4's hand-written block (A/B and cross-check)
    uint64_t* progress;            // [nslots] or null: steps whose bytes are stored (first-quarter bytes of the chunk), ~0 = chunk complete: read by the fused ZRLT / RANK inverse (rank_pipe.hip)
};

// One context of a chunk header (decodeHeader :605-710: alphabet, then the frequencies in groups of 6 / 8 behind their bit width).
// f != nullptr: the frequencies are stored at f[sym] (f zeroed by the caller); f == nullptr: position walk only, the frequency bits are
// skipped a group at a time (their values are checked by whoever parses the context for real). Returns false on an invalid header.
template <typename R>
__device__ static bool knz_ans1_parse_ctx(R& r, uint16_t* f, uint32_t llr, uint32_t scale, int& countOut) {
    uint8_t alpha[256];
    int count = 0;
    if (r.read(1) == 0) {
        if (r.read(1) == 1) count = 0;
        else { count = 256; if (f) for (int i = 0; i < 256; i++) alpha[i] = (uint8_t)i; }
    } else {
        const uint32_t lastMask = r.read(5);
        if (f) {
            for (uint32_t mm = 0; mm <= lastMask; mm++) {
                const uint32_t mask = r.read(8);
                for (int j = 0; j < 8; j++) if ((mask >> j) & 1) alpha[count++] = (uint8_t)(8 * mm + j);
            }
        } else {                                               // position walk only: the symbols themselves are not needed, four masks per read
            for (uint32_t bitsLeft = 8 * (lastMask + 1); bitsLeft > 0;) {
                const uint32_t take = bitsLeft > 32 ? 32 : bitsLeft;
                count += __popc(r.read(take));
                bitsLeft -= take;
            }
        }
    }
    countOut = count;
    if (count == 0) return true;
    const int chk = count < 64 ? 6 : 8;
    uint32_t sum = 0;
    for (int i = 1; i < count; i += chk) {
        const uint32_t logMax = r.read(llr);
        if ((1u << logMax) > scale) return false;
        const int endj = min(i + chk, count);
        if (!f) { r.skip_bits((uint32_t)(endj - i) * logMax); continue; }
        for (int j = i; j < endj; j++) {
            uint32_t fr = 1;
            if (logMax > 0) { fr = 1 + r.read(logMax); if (fr >= scale) return false; }
            f[alpha[j]] = (uint16_t)fr;
            sum += fr;
        }
    }
    if (f) {
        if (scale <= sum) return false;
        f[alpha[0]] = (uint16_t)(scale - sum);
    }
    return true;
}

// Parses the chunk header at reader position r. When freq16 != nullptr the frequencies of context k are stored at freq16[k*256 + sym].
// When ctxpos != nullptr (the walker: every lane runs the same parse, `store` = the lane that writes) the bit position of every context's
// header is left in ctxpos[0..255] and the position behind the last one in ctxpos[256]: the table kernel then parses the 256 contexts side
// by side. Returns false on an invalid header.
template <typename R>
__device__ static bool knz_ans1_parse_header(R& r, uint16_t* freq16, uint32_t& lrOut, int& totalAlpha, uint64_t* ctxpos, bool store) {
    const uint32_t lr = 8 + r.read(3);
    lrOut = lr;
    if (lr > 16) return false;
    uint32_t llr = 3;
    while ((1u << llr) <= lr) llr++;
    const uint32_t scale = 1u << lr;
    totalAlpha = 0;
    for (int k = 0; k < 256; k++) {
        if (ctxpos && store) ctxpos[k] = r.tell();
        int count = 0;
        if (!knz_ans1_parse_ctx(r, freq16 ? freq16 + (size_t)k * 256 : nullptr, llr, scale, count)) return false;
        totalAlpha += count;
    }
    if (ctxpos && store) ctxpos[256] = r.tell();
    return true;
}

// one workgroup per chunk: lane 0 parses the 256 context headers, then all threads fill the slot table
__global__ __launch_bounds__(256) void knz_ans1_dec_tables_kernel(Ans1DecArgs a, uint16_t* freq16, uint16_t* cum16, const uint64_t* ctx_bit) {
    __shared__ int s_mode;
    __shared__ uint32_t s_lr;
    __shared__ int s_total, s_badctx;
    const int tid = threadIdx.x;
    const uint32_t slotId = blockIdx.x;
    const uint32_t b = slotId / a.chunks_per_block, k = slotId % a.chunks_per_block;
    uint32_t* info = a.info + (size_t)slotId * 8;
    const uint32_t preLen = a.blk_pre_len[b];
    if (tid == 0) info[0] = 0;
    if (a.blk_status[b] != 0 || (uint64_t)k * KNZ_ANS1_CHUNK >= preLen) return;
    if ((a.blk_mode[b] & 0x80) || preLen <= 32) { if (tid == 0) info[0] = 1; return; }   // raw
    uint16_t* f16 = freq16 + (size_t)slotId * 65536;
    for (int i = tid; i < 65536; i += 256) f16[i] = 0;
    __syncthreads();
    // the walker left the bit position of every context header (round 3): thread = context parses its own; without them (tests of the
    // single entropy object come through the same walker, so this is the only caller) thread 0 parses all 256 in a row
    const uint64_t* cpos = ctx_bit ? ctx_bit + (size_t)slotId * 257 : nullptr;
    if (cpos) {
        if (tid == 0) { s_total = 0; s_badctx = 0; }
        __syncthreads();
        KnzStreamReader r0;
        r0.init(a.stream, a.nbytes, a.chunk_bit[slotId]);
        const uint32_t lr = 8 + r0.read(3);
        uint32_t llr = 3;
        while ((1u << llr) <= lr) llr++;
        if (lr <= 16) {
            KnzStreamReader r;
            r.init(a.stream, a.nbytes, cpos[tid]);
            int count = 0;
            if (!knz_ans1_parse_ctx(r, f16 + (size_t)tid * 256, llr, 1u << lr, count) || r.tell() != cpos[tid + 1]) s_badctx = 1;
            if (count) atomicAdd(&s_total, count);
        } else if (tid == 0) s_badctx = 1;
        if (tid == 0) s_lr = lr;
        __syncthreads();
    }
    if (tid == 0) {
        KnzStreamReader r;
        r.init(a.stream, a.nbytes, cpos ? cpos[256] : a.chunk_bit[slotId]);
        uint32_t lr = cpos ? s_lr : 0; int total = cpos ? s_total : 0;
        int m = 3;
        if (cpos ? (s_badctx != 0 || total == 0 || lr != KNZ_ANS1_LR)
                 : (!knz_ans1_parse_header(r, f16, lr, total, nullptr, false) || total == 0 || lr != KNZ_ANS1_LR)) m = -1;   // encoder only produces lr 11
        else {
            const uint32_t sz = knz_read_varint(r);
            if (sz >= (1u << 27)) m = -1;
            for (int c = 0; c < 4; c++) info[1 + c] = r.read(32);
            a.paybit[slotId] = r.tell();
            if (r.tell() + 8ull * sz > (a.nbytes << 3) + 7) m = -1;
        }
        s_mode = m; s_lr = lr;
        info[0] = (uint32_t)m; info[5] = lr;
        if (m < 0) a.blk_status[b] = KNZ_ERR_PROCESS_BLOCK;
    }
    __syncthreads();
    if (s_mode != 3) return;
    // thread = context: cumulated frequencies (for the LDS decoder: cum[ctx][0..256], 2048 behind the last symbol)
    // and the slot table of that context (for the table decoder)
    uint32_t* dt = a.dtab + ((size_t)slotId * 256 + tid) * KNZ_ANS1_SCALE;
    const uint16_t* f = f16 + (size_t)tid * 256;
    if (cum16) {
        uint16_t* cq = cum16 + ((size_t)slotId * 256 + tid) * KNZ_ANS1_CUM_STRIDE;
        uint32_t cc = 0;
        for (uint32_t s = 0; s < 256; s++) { cq[s] = (uint16_t)cc; cc += f[s]; }
        cq[256] = (uint16_t)cc;
        return;
    }
    uint32_t cum = 0;
    for (uint32_t s = 0; s < 256; s++) {
        const uint32_t fr = f[s];
        if (fr == 0) continue;
        const uint32_t fclamp = fr < (KNZ_ANS1_SCALE - 1) ? fr : (KNZ_ANS1_SCALE - 1);        // decSymbol.reset :972-977
        for (uint32_t j = 0; j < fr && cum + j < KNZ_ANS1_SCALE; j++) dt[cum + j] = s | (fclamp << 8) | (j << 20);
        cum += fr;
    }
}

__global__ __launch_bounds__(64) void knz_ans1_decode_kernel(Ans1DecArgs a) {
    const int lane = threadIdx.x;
    const int c = lane & 3;
    const uint32_t slotId = blockIdx.x * 16 + (lane >> 2);
    const uint64_t limit = a.nbytes << 3;
    bool live = false;
    uint32_t n = 0, b = 0, k = 0;
    if (slotId < a.nslots) {
        b = slotId / a.chunks_per_block; k = slotId % a.chunks_per_block;
        const uint32_t preLen = a.blk_pre_len[b];
        if (a.info[(size_t)slotId * 8] == 3 && a.blk_status[b] == 0) { live = true; n = min(KNZ_ANS1_CHUNK, preLen - k * KNZ_ANS1_CHUNK); }
    }
    uint8_t* dst = live ? (uint8_t*)a.blk_out_off[b] + (size_t)k * KNZ_ANS1_CHUNK : nullptr;
    const uint32_t* __restrict__ dt = a.dtab + (size_t)(live ? slotId : 0) * 256 * KNZ_ANS1_SCALE;
    const uint64_t paybit = live ? a.paybit[slotId] : 0;
    uint32_t st = live ? a.info[(size_t)slotId * 8 + 1 + c] : 0;
    const uint32_t end4 = n & ~3u;
    const uint32_t q = end4 >> 2;
    const uint32_t steps = live ? q : 0;
    uint32_t maxSteps = steps;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = wave_shfl(maxSteps, lane ^ d); maxSteps = o > maxSteps ? o : maxSteps; }
    uint8_t* qdst = dst + (size_t)c * q;
    uint32_t ctx = 0, cnt = 0;
    const int gshift = (lane >> 2) << 2;
    // decoded bytes are staged per lane in LDS (row stride 132 B: conflict free) and written out every 128 steps, so that
    // no global store sits between two dependent table loads (stores and loads share the in-order vmcnt counter)
    __shared__ uint8_t s_obuf[64][132];
    uint8_t* orow = s_obuf[lane];
    for (uint32_t t = 0; t < maxSteps; t++) {
        const bool act = t < steps;
        uint32_t need = 0;
        if (act) {
            const uint32_t slot = st & (KNZ_ANS1_SCALE - 1);
            const uint32_t e = dt[(ctx << KNZ_ANS1_LR) | slot];
            const uint32_t sym = e & 0xFF;
            orow[t & 127] = (uint8_t)sym;
            st = ((e >> 8) & 0xFFF) * (st >> KNZ_ANS1_LR) + (e >> 20);          // freq*(st>>lr) + (slot - cumFreq)  (:846-858)
            ctx = sym;
            need = st < (1u << 15) ? 1u : 0u;
        }
        const uint64_t bal = wave_ballot(need != 0);
        const uint32_t gb = (uint32_t)(bal >> gshift) & 0xFu;
        if (need) {
            const uint32_t r = cnt + (uint32_t)__popc(gb >> (c + 1));           // refill order st3, st2, st1, st0 (:918-949)
            const uint32_t w = knz_fetch32(a.stream, (int64_t)(paybit + 16ull * r), (int64_t)limit) >> 16;
            st = (st << 16) | w;
        }
        cnt += (uint32_t)__popc(gb);
        if ((t & 127) == 127 || t + 1 == maxSteps) {
            const uint32_t base = t & ~127u;
            if (base < steps) {
                const uint32_t m = min(128u, steps - base);
                for (uint32_t i = 0; i < m; i++) qdst[base + i] = orow[i];
            }
        }
    }
    if (live && c == 0) {
        for (uint32_t i = end4; i < n; i++)
            dst[i] = (uint8_t)(knz_fetch32(a.stream, (int64_t)(paybit + 16ull * cnt + 8ull * (i - end4)), (int64_t)limit) >> 24);
    }
}

#ifdef KNZ_MEASURE                       // the round-2 loop of the LDS decoder: kept for A/B measurements only (-DKNZ_MEASURE, KNZ_ANS1_LDS1)
// LDS-resident decoder: one wave per chunk. The 2 MiB slot table of the decoder above costs one trip to the MALL per step
// (~0.85 us: 64 K of them do not fit any cache level that is close); here the chunk keeps only the cumulated frequencies of its
// 256 contexts in LDS (257 x u16 each, 129 KiB of the CU's 160 KiB) and finds the symbol of a slot with two 16-way searches:
// lanes 16g..16g+15 serve state g; lane l reads cum[ctx][16 l], a ballot + popcount picks the group of 16 symbols, a second
// read + ballot the symbol; freq = cum[s + 1] - cum[s]. Renormalisation words are staged from the stream into an LDS ring.
__global__ __launch_bounds__(64) void knz_ans1_decode_lds_kernel(Ans1DecArgs a, const uint16_t* cum16) {
    __shared__ uint16_t s_cum[256 * KNZ_ANS1_CUM_STRIDE];
    __shared__ uint16_t s_pay[KNZ_ANS1_PAYRING];
    __shared__ uint8_t s_ob[4][256];
    const int lane = threadIdx.x;
    const int g = lane >> 4, l = lane & 15;
    const uint32_t slotId = blockIdx.x;
    const uint64_t limit = a.nbytes << 3;
    const uint32_t b = slotId / a.chunks_per_block, k = slotId % a.chunks_per_block;
    if (a.info[(size_t)slotId * 8] != 3 || a.blk_status[b] != 0) return;
    const uint32_t preLen = a.blk_pre_len[b];
    const uint32_t n = min(KNZ_ANS1_CHUNK, preLen - k * KNZ_ANS1_CHUNK);
    uint8_t* dst = (uint8_t*)a.blk_out_off[b] + (size_t)k * KNZ_ANS1_CHUNK;
    const uint64_t paybit = a.paybit[slotId];
    {
        const uint32_t* src = (const uint32_t*)(cum16 + (size_t)slotId * 256 * KNZ_ANS1_CUM_STRIDE);   // 131,584 B, 4-byte aligned
        for (uint32_t i = lane; i < 256 * KNZ_ANS1_CUM_STRIDE / 2; i += 64) ((uint32_t*)s_cum)[i] = src[i];
    }
    uint32_t st = a.info[(size_t)slotId * 8 + 1 + g];
    const uint32_t end4 = n & ~3u;
    const uint32_t q = end4 >> 2;
    uint8_t* qdst = dst + (size_t)g * q;
    uint32_t ctx = 0, cnt = 0;
    uint32_t payHi = 0;                                                  // words [payHi - ring, payHi) are staged
    const uint64_t below = ((uint64_t)1 << lane) - 1;
    wave_sync();
    for (uint32_t t = 0; t < q; t++) {
        if (cnt + 4 > payHi) {                                           // stage the next half ring of renormalisation words
            wave_sync();
            for (uint32_t j = lane; j < KNZ_ANS1_PAYRING / 2; j += 64) {
                const uint32_t wi = payHi + j;
                s_pay[wi & (KNZ_ANS1_PAYRING - 1)] = (uint16_t)(knz_fetch32(a.stream, (int64_t)(paybit + 16ull * wi), (int64_t)limit) >> 16);
            }
            payHi += KNZ_ANS1_PAYRING / 2;
            wave_sync();
        }
        const uint32_t slot = st & (KNZ_ANS1_SCALE - 1);
        const uint16_t* cq = s_cum + ctx * KNZ_ANS1_CUM_STRIDE;
        // 16 x 16 search: largest s with cum[s] <= slot (an absent symbol shares its cum with the next present one)
        const uint32_t mA = (uint32_t)(wave_ballot(cq[16 * l] <= slot) >> (16 * g)) & 0xFFFFu;
        const uint32_t gi = (uint32_t)__popc(mA) - 1;
        const uint32_t mB = (uint32_t)(wave_ballot(cq[16 * gi + l] <= slot) >> (16 * g)) & 0xFFFFu;
        const uint32_t sym = 16 * gi + (uint32_t)__popc(mB) - 1;
        const uint32_t lo = cq[sym], hi = cq[sym + 1];
        const uint32_t fr = min(hi - lo, (uint32_t)KNZ_ANS1_SCALE - 1);     // decSymbol.reset :972-977
        if (l == 0) s_ob[g][t & 255] = (uint8_t)sym;
        st = fr * (st >> KNZ_ANS1_LR) + slot - lo;                          // (:846-858)
        ctx = sym;
        const bool need = st < (1u << 15);
        const uint64_t nb = wave_ballot(need);
        const uint32_t gb = (uint32_t)((nb & 1) | ((nb >> 15) & 2) | ((nb >> 30) & 4) | ((nb >> 45) & 8));
        if (need) {
            const uint32_t r = cnt + (uint32_t)__popc(gb >> (g + 1));          // refill order st3, st2, st1, st0 (:918-949)
            st = (st << 16) | s_pay[r & (KNZ_ANS1_PAYRING - 1)];
        }
        cnt += (uint32_t)__popc(gb);
        if ((t & 255) == 255 || t + 1 == q) {
            wave_sync();
            const uint32_t base = t & ~255u, m = t + 1 - base;
            for (uint32_t i = l; i < m; i += 16) qdst[base + i] = s_ob[g][i];
            wave_sync();
        }
    }
    if (lane == 0) {
        for (uint32_t i = end4; i < n; i++)
            dst[i] = (uint8_t)(knz_fetch32(a.stream, (int64_t)(paybit + 16ull * cnt + 8ull * (i - end4)), (int64_t)limit) >> 24);
    }
}
#endif

// The same decoder with the per-step bookkeeping taken off the chain (round 3). The loop above spends 75 instructions per step of
// 4 symbols, of which the two 16-way searches and the state update are ~40: the rest was (a) ten scalar instructions that pull the
// four "needs a word" bits out of the ballot (each one a VALU -> SALU hand-over), (b) a predicated byte store per symbol, (c) the
// "ring low?" and "256 symbols decoded?" tests on every step. Here: tiles of 256 steps with the ring topped up and the output
// flushed between tiles only; the rank of a state among the states that refill in the same step (refill order st3, st2, st1, st0,
// :918-949) is a popcount of the ballot under per-lane masks (vector ALU only); four decoded symbols are collected in a register
// and leave as one LDS dword; the renormalisation word is read unconditionally and selected.
#define KNZ_ANS1_PAYRING2 4096                        // words: a tile of 256 steps consumes at most 1024
__global__ __launch_bounds__(64) void knz_ans1_decode_lds2_kernel(Ans1DecArgs a, const uint16_t* cum16) {
    __shared__ uint16_t s_cum[256 * KNZ_ANS1_CUM_STRIDE];
    __shared__ uint16_t s_pay[KNZ_ANS1_PAYRING2 + 4];                    // + the first four words again behind the end: a window of four never wraps
    __shared__ uint32_t s_ob[4][64];
    __shared__ uint16_t s_l1[256 * 16];                                  // first level of the search, cum[ctx][16 l] side by side: the 16 lanes of a state read 32 consecutive bytes (in s_cum they are 32 bytes apart: four lanes to a bank)
    const int lane = threadIdx.x;
    const int grp = lane >> 4, l = lane & 15;
    const int g = 3 - grp;                                               // state 3 in lanes 0..15: the states that refill BEFORE a state (the higher ones, :918-949) sit in the lanes below it
    const uint32_t slotId = blockIdx.x;
    const uint64_t limit = a.nbytes << 3;
    const uint32_t b = slotId / a.chunks_per_block, k = slotId % a.chunks_per_block;
    if (a.info[(size_t)slotId * 8] != 3 || a.blk_status[b] != 0) return;
    const uint32_t preLen = a.blk_pre_len[b];
    const uint32_t n = min(KNZ_ANS1_CHUNK, preLen - k * KNZ_ANS1_CHUNK);
    uint8_t* dst = (uint8_t*)a.blk_out_off[b] + (size_t)k * KNZ_ANS1_CHUNK;
    const uint64_t paybit = a.paybit[slotId];
    {
        const uint32_t* src = (const uint32_t*)(cum16 + (size_t)slotId * 256 * KNZ_ANS1_CUM_STRIDE);   // 131,584 B, 4-byte aligned
        for (uint32_t i = lane; i < 256 * KNZ_ANS1_CUM_STRIDE / 2; i += 64) ((uint32_t*)s_cum)[i] = src[i];
        wave_sync();
        for (uint32_t i = lane; i < 256 * 16; i += 64) s_l1[i] = s_cum[(i >> 4) * KNZ_ANS1_CUM_STRIDE + 16 * (i & 15)];
    }
    uint32_t st = a.info[(size_t)slotId * 8 + 1 + g];
    const uint32_t end4 = n & ~3u;
    const uint32_t q = end4 >> 2;
    uint8_t* qdst = dst + (size_t)g * q;
    uint32_t ctx = 0, cnt = 0;
    uint32_t payHi = 0;                                                  // words [payHi - ring, payHi) are staged
    // ballot bits of the states that refill BEFORE this one in a step (the higher ones), and of all four states (lane 16 g' speaks for state g')
    const uint64_t allm = 0x0001000100010001ull;
    const uint64_t higher = allm & (((uint64_t)1 << (16 * grp)) - 1);   // lane 16 k speaks for the state in lanes 16 k .. 16 k + 15
    const uint32_t hLo = (uint32_t)higher, hHi = (uint32_t)(higher >> 32);
    const uint32_t sh = 16u * (uint32_t)grp;
    wave_sync();
    for (uint32_t t0 = 0; t0 < q; t0 += 256) {
        const uint32_t tn = min(256u, q - t0);
        while (cnt + 1024 + 4 > payHi) {                                 // top the ring up (a quarter at a time: what is still unread stays)
            wave_sync();
            for (uint32_t j = lane; j < KNZ_ANS1_PAYRING2 / 4; j += 64) {
                const uint32_t wi = payHi + j;
                const uint16_t wv = (uint16_t)(knz_fetch32(a.stream, (int64_t)(paybit + 16ull * wi), (int64_t)limit) >> 16);
                const uint32_t at = wi & (KNZ_ANS1_PAYRING2 - 1);
                s_pay[at] = wv;
                if (at < 4) s_pay[KNZ_ANS1_PAYRING2 + at] = wv;
            }
            payHi += KNZ_ANS1_PAYRING2 / 4;
            wave_sync();
        }
        // the (at most four) renormalisation words of a step are taken from a 64-bit window that is read from the ring as soon as the
        // step before has counted its words: the LDS round trip of the word is off the path from one state to the next
        auto load_win = [&](uint32_t c2) -> uint64_t {                       // c2 = 2 x the number of words consumed
            const uint16_t* wp = (const uint16_t*)((const uint8_t*)s_pay + (c2 & (2 * KNZ_ANS1_PAYRING2 - 2)));
            return (uint64_t)wp[0] | ((uint64_t)wp[1] << 16) | ((uint64_t)wp[2] << 32) | ((uint64_t)wp[3] << 48);
        };
        uint32_t cnt2 = 2 * cnt;
        uint64_t win = load_win(cnt2);
        uint32_t c1 = s_l1[16 * ctx + l];                                  // first level of the search for the context of the coming step
        uint32_t acc = 0;
        // one step of the four states; SHIFT = where the symbol goes in the collected word. The true lanes of a compare are a PREFIX of the
        // state's 16 lanes (cum is sorted), so "how many" is 32 - (leading zeros of the 16-bit group mask): v_ffbh on a word operand, one
        // instruction instead of mask + popcount.
        auto step = [&](uint32_t shift) {
            const uint32_t slot = st & (KNZ_ANS1_SCALE - 1);
            const uint16_t* cq = s_cum + knz_mul24(ctx, KNZ_ANS1_CUM_STRIDE);
            // 16 x 16 search: largest s with cum[s] <= slot (an absent symbol shares its cum with the next present one)
            const uint32_t mA = (uint32_t)(wave_ballot(c1 <= slot) >> sh) & 0xFFFFu;
            const uint32_t gi31 = 527u - 16u * (uint32_t)__builtin_clz(mA);                       // 16 (popc - 1) + 31
            const uint32_t mB = (uint32_t)(wave_ballot(cq[gi31 - 31u + l] <= slot) >> sh) & 0xFFFFu;
            const uint32_t sym = gi31 - (uint32_t)__builtin_clz(mB);                              // 16 gi + popc - 1
            const uint32_t lo = cq[sym], hi = cq[sym + 1];
            c1 = s_l1[16 * sym + l];
            const uint32_t fr = min(hi - lo, (uint32_t)KNZ_ANS1_SCALE - 1);     // decSymbol.reset :972-977
            st = fr * (st >> KNZ_ANS1_LR) + slot - lo;                          // (:846-858)
            ctx = sym;
            acc |= sym << shift;
            const bool need = st < (1u << 15);
            const uint64_t nb = wave_ballot(need);
            const uint32_t nlo = (uint32_t)nb, nhi = (uint32_t)(nb >> 32);
            const uint32_t rk = (uint32_t)__popc(nlo & hLo) + (uint32_t)__popc(nhi & hHi);        // refill order st3, st2, st1, st0 (:918-949): 0..3
            const uint32_t w = wave_in_vgpr((uint32_t)(win >> (16u * rk)));                         // (low half) taken whether needed or not: no branch around it
            st = need ? knz_byte_perm(st, w, 0x05040100u) : st;                 // (st << 16) | (w & 0xFFFF)
            cnt2 += (uint32_t)__popcll(nb) >> 3;                                // all 16 lanes of a state speak: one scalar popcount
            win = load_win(cnt2);
        };
        const uint32_t t4 = tn >> 2;
#ifndef KNZ_HIP_EMU
        if (t4 != 0 && a.plain_loop == 0) {
            // The four steps per output word as ONE hand-written block with its own loop (a lone wave pays for every instruction it issues).
            // Round 5: ONE LDS round trip on the path from a state to the next instead of two. The second level of the search is read as
            // (cum[k], cum[k + 1]) per lane (two reads in flight together), every lane of the state's sixteen computes the new state from ITS
            // pair, the one lane whose pair holds the slot keeps it (lane l with l + ffbh(mask) == 31) and four rotate-and-or moves inside the
            // row of sixteen (DPP row_ror 1 / 2 / 4 / 8) hand it to the others: the read of cum[sym], cum[sym + 1] that waited for the symbol is
            // gone. The slots behind a compare whose mask is read as data and behind a write that a DPP move reads hold the next things the
            // step needs (the first-level read of the coming step, the collected word), s_nop only where nothing is left.
            // v254:v255 = the 64-bit scratch of the two mask shifts and of the window shift (named, so that its halves can be addressed).
#define KNZ_A1_STEP(SHIFT, EXTRA) \
            "v_and_b32_e32 %[slot], 0x7ff, %[st]\n\t" \
            "s_waitcnt lgkmcnt(0)\n\t"                     /* the first-level value */ \
            "v_cmp_ge_u32_e32 vcc, %[slot], %[c1]\n\t" \
            "v_mad_u32_u24 %[basel], %[ctx], %[k514], %[cumL]\n\t"      /* &cum[ctx][16 * 31 + l] */ \
            "s_add_u32 %[cnt], %[cnt], %[wds]\n\t"                       /* the words the step before took */ \
            "v_lshrrev_b64 v[254:255], %[sh], vcc\n\t" \
            "v_ffbh_u32_sdwa v254, v254 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" \
            "v_mad_i32_i24 %[adr], v254, %[m32], %[basel]\n\t"          /* group 31 - clz: 32 bytes per group */ \
            "ds_read_u16 %[c2], %[adr]\n\t"                              /* cum[k], k = 16 group + l */ \
            "ds_read_u16 %[c2h], %[adr] offset:2\n\t"                    /* cum[k + 1] */ \
            "s_and_b32 %[s0], %[cnt], 0xfff\n\t"                         /* under that round trip: this step's renormalisation window ... */ \
            "v_lshl_add_u32 %[adr], %[s0], 1, %[payA]\n\t" \
            "ds_read_b64 %[win], %[adr]\n\t" \
            "v_lshrrev_b32_e32 %[hi11], 11, %[st]\n\t"                   /* ... and what else does not wait for the pair */ \
            "v_mad_i32_i24 %[g16], v254, -16, %[c527]\n\t"              /* 16 (31 - clz) + 31 */ \
            "s_waitcnt lgkmcnt(1)\n\t"                                   /* the pair (the window behind it may still be on its way) */ \
            "v_cmp_ge_u32_e32 vcc, %[slot], %[c2]\n\t" \
            "v_sub_u32_e32 %[fr], %[c2h], %[c2]\n\t"                     /* this lane's frequency ... */ \
            "v_sub_u32_e32 %[slot], %[slot], %[c2]\n\t"                  /* ... and slot - cum[k] */ \
            "v_lshrrev_b64 v[254:255], %[sh], vcc\n\t" \
            "v_ffbh_u32_sdwa v254, v254 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" \
            "v_sub_u32_e32 %[ctx], %[g16], v254\n\t"                    /* the symbol = the next context */ \
            "v_cmp_eq_u32_e32 vcc, %[l31], v254\n\t"                     /* the lane whose pair holds the slot: l = popc - 1 = 31 - ffbh */ \
            "v_min_u32_e32 %[fr], 0x7ff, %[fr]\n\t"                      /* decSymbol.reset :972-977 */ \
            "v_mad_u32_u24 %[fr], %[fr], %[hi11], %[slot]\n\t"           /* freq (st >> 11) + slot - cum (:846-858), right in that lane only */ \
            "v_cndmask_b32_e32 %[st], 0, %[fr], vcc\n\t" \
            "v_lshl_add_u32 %[adr], %[ctx], 5, %[l1L]\n\t" \
            "ds_read_u16 %[c1], %[adr]\n\t"                              /* first level of the coming step */ \
            "v_or_b32_dpp %[st], %[st], %[st] row_ror:1 row_mask:0xf bank_mask:0xf\n\t" \
            "v_lshl_or_b32 %[acc], %[ctx], " SHIFT ", %[acc]\n\t" \
            EXTRA                                                        /* the second of the two slots in front of the next DPP move */ \
            "v_or_b32_dpp %[st], %[st], %[st] row_ror:2 row_mask:0xf bank_mask:0xf\n\t" \
            "s_nop 1\n\t" \
            "v_or_b32_dpp %[st], %[st], %[st] row_ror:4 row_mask:0xf bank_mask:0xf\n\t" \
            "s_nop 1\n\t" \
            "v_or_b32_dpp %[st], %[st], %[st] row_ror:8 row_mask:0xf bank_mask:0xf\n\t" \
            "v_cmp_gt_u32_e32 vcc, %[thr], %[st]\n\t"                   /* which states renormalise */ \
            "s_and_b64 s[98:99], vcc, %[m15]\n\t"                     /* one bit per state, in the top lane of its sixteen (s98:s99 named: its halves are operands) */ \
            "s_bcnt1_i32_b64 %[wds], s[98:99]\n\t" \
            "v_mbcnt_lo_u32_b32 v254, s98, 0\n\t"                      /* states in the lanes below that renormalise = words in front of this state's */ \
            "v_mbcnt_hi_u32_b32 v254, s99, v254\n\t" \
            "v_lshlrev_b32_e32 v254, 4, v254\n\t" \
            "s_waitcnt lgkmcnt(1)\n\t"                                   /* the window (the coming step's first-level value may still be on its way) */ \
            "v_lshrrev_b64 v[254:255], v254, %[win]\n\t" \
            "v_perm_b32 v254, %[st], v254, %[sel]\n\t" \
            "v_cndmask_b32_e32 %[st], %[st], v254, vcc\n\t"
            uint32_t slot, basel, hi11, g16, adr, c2, c2h, fr, accv, c1v, s0, wds = 0, n4 = wave_uniform(t4), cntw = wave_uniform(cnt2 >> 1);
            uint64_t winv;
            uint32_t obp = knz_lds_addr(&s_ob[g][0]);
            const uint32_t cumA = knz_lds_addr(s_cum), cumL = cumA + 2u * (uint32_t)l + 992u, l1L = knz_lds_addr(s_l1) + 2u * (uint32_t)l, payA = knz_lds_addr(s_pay);
            asm volatile(
                "v_lshl_add_u32 %[adr], %[ctx], 5, %[l1L]\n\t"
                "ds_read_u16 %[c1], %[adr]\n"
                ".Lknz_a1d_loop_%=:\n\t"
                "v_mov_b32_e32 %[acc], 0\n\t"
                KNZ_A1_STEP("0", "s_nop 0\n\t") KNZ_A1_STEP("8", "s_nop 0\n\t") KNZ_A1_STEP("16", "s_nop 0\n\t")
                KNZ_A1_STEP("24", "ds_write_b32 %[obp], %[acc]\n\t")              /* the collected word leaves in a slot that would be empty */
                "v_add_u32_e32 %[obp], 4, %[obp]\n\t"
                "s_sub_u32 %[n4], %[n4], 1\n\t"
                "s_cmp_lg_u32 %[n4], 0\n\t"
                "s_cbranch_scc1 .Lknz_a1d_loop_%=\n\t"
                "s_add_u32 %[cnt], %[cnt], %[wds]\n\t"
                "s_waitcnt lgkmcnt(0)"
                : [st] "+v"(st), [ctx] "+v"(ctx), [cnt] "+s"(cntw), [obp] "+v"(obp), [n4] "+s"(n4), [wds] "+s"(wds),
                  [slot] "=&v"(slot), [basel] "=&v"(basel), [hi11] "=&v"(hi11), [g16] "=&v"(g16), [adr] "=&v"(adr), [c2] "=&v"(c2), [c2h] "=&v"(c2h), [fr] "=&v"(fr),
                  [acc] "=&v"(accv), [c1] "=&v"(c1v), [win] "=&v"(winv), [s0] "=&s"(s0)
                : [sh] "v"(sh), [cumL] "v"(cumL), [l1L] "v"(l1L), [payA] "v"(payA), [l31] "v"(31u - (uint32_t)l), [thr] "s"(1u << 15), [sel] "s"(0x05040100u),
                  [k514] "s"(2u * KNZ_ANS1_CUM_STRIDE), [m32] "s"(-32), [c527] "s"(527u), [m15] "s"(0x8000800080008000ull)
                : "vcc", "scc", "v254", "v255", "s98", "s99", "memory");
            cnt2 = 2 * cntw;
#undef KNZ_A1_STEP
            c1 = s_l1[16 * ctx + l];
            win = load_win(cnt2);
            acc = 0;
        } else
        if (t4 != 0 && a.plain_loop == 2) {
            // (round 3-4's block, kept for A/B and cross-check: KNZ_ANS1_LOHI_LDS. Two LDS round trips per step: the second level, then cum[sym], cum[sym + 1])
            // The same four steps per output word as ONE hand-written block with its own loop (a lone wave pays for every instruction it
            // issues: no s_nop - the two slots behind each compare whose mask is read as data hold the next things the step needs - and
            // three instructions less address arithmetic per step than the compiler's schedule of the lambda above). LDS reads in flight
            // never cross the block's boundary: it starts with the two reads the first step waits for and ends with a wait.
            // v254:v255 = the 64-bit scratch of the two mask shifts and of the window shift (named, so that its halves can be addressed).
#define KNZ_A1_STEP(SHIFT, EXTRA) \
            "v_and_b32_e32 %[slot], 0x7ff, %[st]\n\t" \
            "v_mad_u32_u24 %[basel], %[ctx], %[k514], %[cumL]\n\t"      /* &cum[ctx][16 * 31 + l] */ \
            "s_waitcnt lgkmcnt(1)\n\t"                     /* the first-level value; the window (issued after it) may still be on its way */ \
            "v_cmp_ge_u32_e32 vcc, %[slot], %[c1]\n\t" \
            "v_lshrrev_b32_e32 %[hi11], 11, %[st]\n\t" \
            "v_mad_u32_u24 %[base], %[ctx], %[k514], %[cumA]\n\t"       /* &cum[ctx][0] */ \
            "v_lshrrev_b64 v[254:255], %[sh], vcc\n\t" \
            "v_ffbh_u32_sdwa v254, v254 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" \
            "v_mad_i32_i24 %[adr], v254, %[m32], %[basel]\n\t"          /* group 31 - clz: 32 bytes per group */ \
            "ds_read_u16 %[c2], %[adr]\n\t" \
            "v_mad_i32_i24 %[g16], v254, -16, %[c527]\n\t"              /* 16 (31 - clz) + 31 */ \
            "s_waitcnt lgkmcnt(0)\n\t" \
            "v_cmp_ge_u32_e32 vcc, %[slot], %[c2]\n\t" \
            "s_nop 1\n\t" \
            "v_lshrrev_b64 v[254:255], %[sh], vcc\n\t" \
            "v_ffbh_u32_sdwa v254, v254 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n\t" \
            "v_sub_u32_e32 %[ctx], %[g16], v254\n\t"                    /* the symbol = the next context */ \
            "v_lshl_add_u32 %[adr], %[ctx], 1, %[base]\n\t" \
            "ds_read_b32 %[lohi], %[adr]\n\t" \
            "v_lshl_add_u32 %[adr], %[ctx], 5, %[l1L]\n\t" \
            "ds_read_u16 %[c1], %[adr]\n\t" \
            "v_lshl_or_b32 %[acc], %[ctx], " SHIFT ", %[acc]\n\t" \
            EXTRA \
            "s_waitcnt lgkmcnt(1)\n\t" \
            "v_sub_u32_sdwa %[g16], %[lohi], %[lohi] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n\t" \
            "v_sub_u32_sdwa %[slot], %[slot], %[lohi] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t" \
            "v_min_u32_e32 %[g16], 0x7ff, %[g16]\n\t" \
            "v_mad_u32_u24 %[st], %[g16], %[hi11], %[slot]\n\t" \
            "v_cmp_gt_u32_e32 vcc, %[thr], %[st]\n\t"                   /* which states renormalise */ \
            "s_and_b64 s[98:99], vcc, %[m15]\n\t"                     /* one bit per state, in the top lane of its sixteen (s98:s99 named: its halves are operands) */ \
            "s_bcnt1_i32_b64 %[s0], s[98:99]\n\t" \
            "v_mbcnt_lo_u32_b32 v254, s98, 0\n\t"                      /* states in the lanes below that renormalise = words in front of this state's */ \
            "v_mbcnt_hi_u32_b32 v254, s99, v254\n\t" \
            "v_lshlrev_b32_e32 v254, 4, v254\n\t" \
            "v_lshrrev_b64 v[254:255], v254, %[win]\n\t" \
            "s_add_u32 %[cnt], %[cnt], %[s0]\n\t" \
            "v_perm_b32 v254, %[st], v254, %[sel]\n\t" \
            "v_cndmask_b32_e32 %[st], %[st], v254, vcc\n\t" \
            "s_and_b32 %[s0], %[cnt], 0xfff\n\t" \
            "v_lshl_add_u32 %[adr], %[s0], 1, %[payA]\n\t" \
            "ds_read_b64 %[win], %[adr]\n\t"
            uint32_t slot, base, basel, hi11, g16, adr, c2, lohi, accv, c1v, s0, n4 = wave_uniform(t4), cntw = wave_uniform(cnt2 >> 1);
            uint64_t winv;
            uint32_t obp = knz_lds_addr(&s_ob[g][0]);
            const uint32_t cumA = knz_lds_addr(s_cum), cumL = cumA + 2u * (uint32_t)l + 992u, l1L = knz_lds_addr(s_l1) + 2u * (uint32_t)l, payA = knz_lds_addr(s_pay);
            asm volatile(
                "v_lshl_add_u32 %[adr], %[ctx], 5, %[l1L]\n\t"
                "ds_read_u16 %[c1], %[adr]\n\t"
                "s_and_b32 %[s0], %[cnt], 0xfff\n\t"
                "v_lshl_add_u32 %[adr], %[s0], 1, %[payA]\n\t"
                "ds_read_b64 %[win], %[adr]\n"
                ".Lknz_a1_loop_%=:\n\t"
                "v_mov_b32_e32 %[acc], 0\n\t"
                KNZ_A1_STEP("0", "") KNZ_A1_STEP("8", "") KNZ_A1_STEP("16", "")
                KNZ_A1_STEP("24", "ds_write_b32 %[obp], %[acc]\n\t")              /* the collected word leaves in front of the step's last two reads' waits */
                "v_add_u32_e32 %[obp], 4, %[obp]\n\t"
                "s_sub_u32 %[n4], %[n4], 1\n\t"
                "s_cmp_lg_u32 %[n4], 0\n\t"
                "s_cbranch_scc1 .Lknz_a1_loop_%=\n\t"
                "s_waitcnt lgkmcnt(0)"
                : [st] "+v"(st), [ctx] "+v"(ctx), [cnt] "+s"(cntw), [obp] "+v"(obp), [n4] "+s"(n4),
                  [slot] "=&v"(slot), [base] "=&v"(base), [basel] "=&v"(basel), [hi11] "=&v"(hi11), [g16] "=&v"(g16), [adr] "=&v"(adr), [c2] "=&v"(c2), [lohi] "=&v"(lohi),
                  [acc] "=&v"(accv), [c1] "=&v"(c1v), [win] "=&v"(winv), [s0] "=&s"(s0)
                : [sh] "v"(sh), [cumL] "v"(cumL), [cumA] "v"(cumA), [l1L] "v"(l1L), [payA] "v"(payA), [thr] "s"(1u << 15), [sel] "s"(0x05040100u),
                  [k514] "s"(2u * KNZ_ANS1_CUM_STRIDE), [m32] "s"(-32), [c527] "s"(527u), [m15] "s"(0x8000800080008000ull)
                : "vcc", "scc", "v254", "v255", "s98", "s99", "memory");
            cnt2 = 2 * cntw;
#undef KNZ_A1_STEP
            c1 = s_l1[16 * ctx + l];
            win = load_win(cnt2);
            acc = 0;
        } else
#endif
        for (uint32_t tq = 0; tq < t4; tq++) {
            step(0); step(8); step(16); step(24);
            s_ob[g][tq] = acc;                                                  // (all 16 lanes of the state hold the same word)
            acc = 0;
        }
        for (uint32_t t = 4 * t4; t < tn; t++) step(8 * (t & 3));
        if (tn & 3) s_ob[g][tn >> 2] = acc;
        cnt = cnt2 >> 1;
        wave_sync();
        {
            const uint8_t* ob = (const uint8_t*)s_ob[g];
            for (uint32_t i = l; i < tn; i += 16) qdst[t0 + i] = ob[i];
        }
        if (a.progress != nullptr && ((t0 >> 8) & 3u) == 3u) {              // every fourth tile (1024 steps, ~0.2 ms): what is stored so far is handed on
            agent_fence_release();
            if (lane == 0) knz_publish64(a.progress + slotId, (uint64_t)(t0 + tn));
        }
        wave_sync();
    }
    if (lane == 0) {
        for (uint32_t i = end4; i < n; i++)
            dst[i] = (uint8_t)(knz_fetch32(a.stream, (int64_t)(paybit + 16ull * cnt + 8ull * (i - end4)), (int64_t)limit) >> 24);
    }
    if (a.progress != nullptr) {
        agent_fence_release();
        if (lane == 0) knz_publish64(a.progress + slotId, ~0ull);
    }
}

// raw chunks of an ANS1 block (copy blocks, <= 32 byte inputs)
__global__ __launch_bounds__(256) void knz_ans1_raw_kernel(Ans1DecArgs a) {
    const uint32_t slotId = blockIdx.x;
    if (a.info[(size_t)slotId * 8] != 1) return;
    const uint32_t b = slotId / a.chunks_per_block, k = slotId % a.chunks_per_block;
    const uint32_t preLen = a.blk_pre_len[b];
    const uint32_t n = min(KNZ_ANS1_CHUNK, preLen - k * KNZ_ANS1_CHUNK);
    uint8_t* dst = (uint8_t*)a.blk_out_off[b] + (size_t)k * KNZ_ANS1_CHUNK;
    const uint64_t cbit = a.chunk_bit[slotId];
    for (uint32_t i = threadIdx.x * 4; i < n; i += 1024) {
        const uint32_t w = knz_fetch32(a.stream, (int64_t)(cbit + 8ull * i), (int64_t)(a.nbytes << 3));
        for (uint32_t j = 0; j < 4 && i + j < n; j++) dst[i + j] = (uint8_t)(w >> (24 - 8 * j));
    }
}
