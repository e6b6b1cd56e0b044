// Decode side of the v6 stream for the Huffman path, gfx950.
//   knz_dec_walk_stream_kernel  : Reader.processBlock's sequential part (v2/io/CompressedStream.go:1816-1852):
//                                 reads (lw-3):5, written:lw of every block until the 0-length end marker.
//   knz_dec_walk_blocks_kernel  : block header (:1878-1914) + the sequential location of every 16 KiB chunk of a
//                                 Huffman payload (HuffmanDecoder.decodeV6 / readLengths / decodeChunkV6,
//                                 v2/entropy/HuffmanCodec.go:758-805,620-657,807-830): chunks start where the
//                                 previous one ended, fragment sizes are explicit varints.
//   knz_huf_decode_kernel       : one wave64 per chunk: rebuild canonical codes + the 4096-entry table
//                                 (buildDecodingTable :661-697) in LDS, then the 4 fragment bit strings are decoded
//                                 by 4 lanes (decodeChunkV6 :832-969).
#include "bits.h"

// MSB-first reader over a 4-byte aligned global buffer, 64-bit window, aligned 32-bit refills.
struct KnzStreamReader {
    const uint32_t* words;
    uint64_t nwords;      // readable words (reads beyond return 0)
    uint64_t next;        // next word index to load
    uint64_t win;         // left aligned
    uint32_t navail;      // valid bits in win
    uint32_t pre;         // word `next` as loaded (little endian), fetched one refill ahead. It is byte-swapped only when it
                          // enters the window: touching it earlier would put the s_waitcnt right behind the load.
    __device__ __forceinline__ uint32_t ld_raw(uint64_t i) const { return i < nwords ? words[i] : 0u; }
    __device__ __forceinline__ uint32_t ld(uint64_t i) const { return knz_bswap32(ld_raw(i)); }
    __device__ __forceinline__ void init(const uint8_t* base, uint64_t nbytes, uint64_t bitpos) {
        words = (const uint32_t*)base;
        nwords = (nbytes + 3) >> 2;
        uint64_t q = bitpos >> 5;
        uint32_t off = (uint32_t)(bitpos & 31);
        win = ((uint64_t)ld(q) << 32) | ld(q + 1);
        win <<= off;
        navail = 64 - off;
        next = q + 2;
        pre = ld_raw(next);
    }
    __device__ __forceinline__ void refill() {
        if (navail <= 32) { win |= (uint64_t)knz_bswap32(pre) << (32 - navail); navail += 32; next++; pre = ld_raw(next); }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) { refill(); return (uint32_t)(win >> (64 - n)); } // 1..32
    __device__ __forceinline__ void skip(uint32_t n) { refill(); win <<= n; navail -= n; }                 // 0..32
    __device__ __forceinline__ void consume(uint32_t n) { win <<= n; navail -= n; }                        // right after peek(>= n)
    __device__ __forceinline__ void skip_bits(uint32_t n) { if (n <= 32) skip(n); else seek(tell() + n); }
    __device__ __forceinline__ uint32_t read(uint32_t n) { uint32_t v = peek(n); win <<= n; navail -= n; return v; }
    __device__ __forceinline__ uint64_t tell() const { return (next << 5) - navail; }
    __device__ __forceinline__ void seek(uint64_t bitpos) { init((const uint8_t*)words, nwords << 2, bitpos); }
};

// Same interface, but the stream words are staged 128 at a time into two registers per lane (two coalesced loads by the
// whole wave) and fetched with v_readlane: a serial header walk then never waits on a dependent global load, and since
// every value is wave-uniform the compiler keeps the parse on the scalar unit. ALL 64 lanes must call it uniformly.
struct KnzWaveReader {
    const uint32_t* words;
    uint64_t nwords;
    uint64_t base;        // word index held by lane 0 of w0
    uint32_t w0, w1;      // lane l: words base+l and base+64+l (byte-swapped on use)
    uint64_t next;
    uint64_t win;
    uint32_t navail;
    __device__ __forceinline__ void stage(uint64_t b) {
        const uint64_t l = (uint64_t)lane_id();
        base = b;
        w0 = (b + l < nwords) ? words[b + l] : 0u;
        w1 = (b + 64 + l < nwords) ? words[b + 64 + l] : 0u;
    }
    __device__ __forceinline__ uint32_t ld(uint64_t i) {
        if (i < base || i >= base + 128) stage(i);
        const uint32_t k = (uint32_t)(i - base);
        const uint32_t v = k < 64 ? wave_readlane(w0, k) : wave_readlane(w1, k - 64);
        return knz_bswap32(v);
    }
    __device__ __forceinline__ void init(const uint8_t* b, uint64_t nbytes, uint64_t bitpos) {
        words = (const uint32_t*)b;
        nwords = (nbytes + 3) >> 2;
        stage(bitpos >> 5);
        seek(bitpos);
    }
    __device__ __forceinline__ void seek(uint64_t bitpos) {
        const uint64_t q = bitpos >> 5;
        const uint32_t off = (uint32_t)(bitpos & 31);
        win = ((uint64_t)ld(q) << 32) | ld(q + 1);
        win <<= off;
        navail = 64 - off;
        next = q + 2;
    }
    __device__ __forceinline__ void refill() {
        if (navail <= 32) { win |= (uint64_t)ld(next++) << (32 - navail); navail += 32; }
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n) { refill(); return (uint32_t)(win >> (64 - n)); }
    __device__ __forceinline__ void skip(uint32_t n) { refill(); win <<= n; navail -= n; }
    // a jump ahead: inside the window when it is short (the header walks jump over six or eight frequencies at a time; seek() rebuilds the window)
    __device__ __forceinline__ void skip_bits(uint32_t n) { if (n <= 32) skip(n); else seek(tell() + n); }
    __device__ __forceinline__ uint32_t read(uint32_t n) { uint32_t v = peek(n); win <<= n; navail -= n; return v; }
    __device__ __forceinline__ uint64_t tell() const { return (next << 5) - navail; }
};

// EntropyUtils.go:278-296
template <typename R>
__device__ __forceinline__ uint32_t knz_read_varint(R& r) {
    uint32_t res = 0, shift = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t v = r.read(8);
        res |= (v & 0x7F) << shift;
        if (v < 128) return res;
        shift += 7;
    }
    uint32_t v = r.read(8);
    return res | ((v & 0x0F) << 28);
}

template <typename R>
__device__ static bool knz_ans1_parse_header(R& r, uint16_t* freq16, uint32_t& lrOut, int& totalAlpha, uint64_t* ctxpos, bool store);

struct WalkStreamArgs {
    const uint8_t* stream; uint64_t nbytes;
    uint64_t first_bit;          // first block's framing
    uint64_t seg_bits;           // != 0: a rank segment of exactly seg_bits bits without end marker (multi-GPU)
    uint32_t max_blocks;
    uint64_t* blk_bit;           // [max_blocks] bit position of the block-local stream
    uint64_t* blk_bits;          // [max_blocks] its length in bits
    uint32_t* result;            // [2] {nblocks, error code}
};

__global__ void knz_dec_walk_stream_kernel(WalkStreamArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    KnzStreamReader r;
    r.init(a.stream, a.nbytes, a.first_bit);
    const uint64_t limit = a.nbytes << 3;
    uint32_t n = 0, err = 0;
    for (;;) {
        if (a.seg_bits && r.tell() >= a.first_bit + a.seg_bits) break;
        if (r.tell() + 8 > limit) { err = KNZ_ERR_PROCESS_BLOCK; break; }     // bitstream EOS panic (:1778-1786)
        uint32_t lr = r.read(5) + 3;
        uint64_t read = 0;
        if (lr > 32) { read = (uint64_t)r.read(lr - 32) << 32; read |= r.read(32); }
        else read = r.read(lr);
        if (read == 0) break;
        if (read > ((uint64_t)1 << 34)) { err = KNZ_ERR_BLOCK_SIZE; break; }
        uint64_t pos = r.tell();
        if (pos + read > limit) { err = KNZ_ERR_PROCESS_BLOCK; break; }
        if (n >= a.max_blocks) { err = KNZ_ERR_BLOCK_SIZE; break; }
        a.blk_bit[n] = pos;
        a.blk_bits[n] = read;
        n++;
        r.seek(pos + read);
    }
    a.result[0] = n;
    a.result[1] = err;
}

struct WalkBlocksArgs {
    const uint8_t* stream; uint64_t nbytes;
    const uint64_t* blk_bit; const uint64_t* blk_bits;
    uint32_t nblocks;
    uint32_t block_size;          // stream block size (bounds preTransformLength, :1893-1903)
    uint32_t checksum_bits;
    uint32_t entropy;
    uint32_t chunks_per_block;
    uint32_t payload_only;        // 1: no block header, the entropy payload starts at blk_bit[b] and decodes to given_len bytes
    uint32_t given_len;
    // outputs
    uint32_t* blk_pre_len;        // [nblocks] preTransformLength
    uint8_t* blk_mode;            // [nblocks]
    uint8_t* blk_skip;            // [nblocks]
    uint64_t* blk_cksum;          // [nblocks]
    uint64_t* chunk_bit;          // [nblocks*CPB] bit position of each chunk's first bit
    int32_t* blk_status;          // [nblocks]
    uint64_t* blk_end_bit;        // [nblocks] bit position just past the entropy payload
    // knz_dec_block_headers_kernel only, when the blocks decode straight to their place (no transform stage): the output placement
    // and the checks the host would make between the walk and the decode (Reader.processBlock :1707-1710, capacity)
    uint32_t check_out;           // 1: fill out_off and fold the checks into blk_status
    uint64_t* out_off;            // [nblocks] out_base + b * out_stride
    uint64_t out_base, out_stride, out_cap;
    uint32_t stream_block_size;
    uint64_t* ans1_ctx_bit;       // [nblocks * CPB][257] rANS order 1: bit position of every context header of a chunk (+ the end), or null
};

// Skips one signed Exp-Golomb code (ExpGolombCodec.go:159-190): '1' or L zeros, 1, L+1 bits.
template <typename R>
__device__ __forceinline__ void knz_skip_expg(R& r) {
    uint32_t w = r.peek(17);
    if (w & 0x10000) { r.skip(1); return; }
    if (w == 0) { r.skip(17); return; }                     // corrupt input; the walk fails its bounds check later
    uint32_t z = (uint32_t)__builtin_clz(w << 15);         // leading zeros inside the 17-bit window
    uint32_t lg = z & 7;                                    // the reference clamps log2 &= 7
    // consumed: z zeros + the '1' + (lg+1) bits
    r.skip(z + 1);
    r.skip(lg + 1);
}

// 12-bit look-ahead table (LDS) for skipping SEVERAL signed Exp-Golomb codes at once: entry = (codes << 4) | bits for
// the complete codes found in the window (code-length deltas are mostly 0 = '1' or +-1 = 4 bits, so a window usually
// holds 3-8 codes). codes == 0: the first code does not fit -> single-code path. (An 8-bit table held in one VGPR and
// read with v_readlane was measured slower: 5.2 ms vs 4.3 ms for the walk of config 2, fewer codes per step.)
#define KNZ_EXPG_WIN 12
__device__ __forceinline__ uint32_t knz_expg_lut_entry(uint32_t w) {
    uint32_t pos = 0, codes = 0;
    while (pos < KNZ_EXPG_WIN) {
        if ((w >> (KNZ_EXPG_WIN - 1 - pos)) & 1) { codes++; pos++; continue; }
        uint32_t z = 0;
        while (pos + z < KNZ_EXPG_WIN && !((w >> (KNZ_EXPG_WIN - 1 - pos - z)) & 1)) z++;
        if (pos + z >= KNZ_EXPG_WIN) break;              // terminator outside the window
        const uint32_t total = z + 1 + (z & 7) + 1;
        if (pos + total > KNZ_EXPG_WIN) break;
        codes++; pos += total;
    }
    return (codes << 4) | pos;
}

struct HufDecArgs {
    const uint8_t* stream; uint64_t nbytes;
    const uint32_t* blk_pre_len;
    const uint8_t* blk_mode;
    const uint64_t* chunk_bit;
    const uint64_t* blk_out_off;   // [nblocks] byte offset of the block's post-transform data in out
    uint32_t chunks_per_block;
    uint32_t entropy;
    uint8_t* out;
    int32_t* blk_status;
};

// Serial form (4 lanes, count-driven like decodeChunkV6). `only` != nullptr: handle just the chunks the parallel kernel
// (huffman_par.hip) handed back.
__global__ __launch_bounds__(64) void knz_huf_decode_kernel(HufDecArgs a, const uint8_t* only) {
    __shared__ uint16_t s_table[1 << KNZ_HUF_MAXLEN];
    __shared__ uint8_t s_len[256];
    __shared__ uint8_t s_alpha[256];
    __shared__ uint16_t s_C[256];
    __shared__ uint8_t s_symAt[256];
    __shared__ uint32_t s_fragbits[4];
    __shared__ uint64_t s_fragpos[4];
    __shared__ int s_count, s_err;
    __shared__ uint64_t s_tailpos;

    const int lane = threadIdx.x;
    if (only && !only[blockIdx.x]) return;
    const uint32_t cpb = a.chunks_per_block;
    const uint32_t b = blockIdx.x / cpb, k = blockIdx.x % cpb;
    const uint32_t preLen = a.blk_pre_len[b];
    if (a.blk_status[b] != 0) return;
    if ((uint64_t)k * KNZ_HUF_CHUNK >= preLen) return;
    const uint32_t n = min((uint32_t)KNZ_HUF_CHUNK, preLen - k * KNZ_HUF_CHUNK);
    uint8_t* dst = a.out + a.blk_out_off[b] + (size_t)k * KNZ_HUF_CHUNK;
    const uint64_t cbit = a.chunk_bit[blockIdx.x];
    if (cbit >= ~0ull - 1) return;                                     // the walk never reached this chunk (fused walk+decode launch)
    uint32_t entropy = a.entropy;
    if (a.blk_mode[b] & 0x80) entropy = KNZ_E_NONE;

    if (entropy == KNZ_E_NONE || n < 32) {   // raw copy at an arbitrary bit offset
        for (uint32_t i = lane * 4; i < n; i += 256) {
            uint32_t w = knz_fetch32(a.stream, (int64_t)(cbit + 8ull * i), (int64_t)(a.nbytes << 3));
            for (uint32_t j = 0; j < 4 && i + j < n; j++) dst[i + j] = (uint8_t)(w >> (24 - 8 * j));
        }
        return;
    }

    // ---- header: alphabet + code lengths (readLengths :620-657), serial on lane 0 ---------------------------------
    for (int i = lane; i < 256; i += 64) s_len[i] = 0;
    if (lane == 0) s_err = 0;
    wave_sync();
    if (lane == 0) {
        KnzStreamReader r;
        r.init(a.stream, a.nbytes, cbit);
        int count = 0;
        if (r.read(1) == 0) {
            r.read(1);
            count = 256;
            for (int i = 0; i < 256; i++) s_alpha[i] = (uint8_t)i;
        } else {
            uint32_t lastMask = r.read(5);
            for (uint32_t m = 0; m <= lastMask; m++) {
                uint32_t mask = r.read(8);
                for (int j = 0; j < 8; j++) if ((mask >> j) & 1) s_alpha[count++] = (uint8_t)(8 * m + j);
            }
        }
        int curSize = 2;
        for (int i = 0; i < count; i++) {
            // signed Exp-Golomb (ExpGolombCodec.go:159-190)
            int delta;
            if (r.read(1) == 1) delta = 0;
            else {
                uint32_t lg = 1;
                while (r.read(1) == 0) lg++;
                lg &= 7;
                uint32_t val = r.read(lg + 1);
                uint32_t res = (val >> 1) + (1u << lg) - 1u;
                if (val & 1) res = ~res + 1u;
                delta = (int)(int8_t)(uint8_t)res;
            }
            curSize = (int)(int8_t)(curSize + delta);
            if (curSize <= 0 || curSize > KNZ_HUF_MAXLEN) { s_err = KNZ_ERR_PROCESS_BLOCK; break; }
            s_len[s_alpha[i]] = (uint8_t)curSize;
        }
        s_count = count;
        if (count > 1 && s_err == 0) {
            uint64_t fb[4];
            for (int j = 0; j < 4; j++) { s_fragbits[j] = knz_read_varint(r); fb[j] = s_fragbits[j]; }
            uint64_t p = r.tell();
            for (int j = 0; j < 4; j++) { s_fragpos[j] = p; p += fb[j]; }
            s_tailpos = p;
        }
    }
    wave_sync();
    if (s_err) { if (lane == 0) a.blk_status[b] = s_err; return; }
    const int count = s_count;
    if (count == 1) {                          // decodeV6 :778-786
        const uint8_t v = s_alpha[0];
        for (uint32_t i = lane; i < n; i += 64) dst[i] = v;
        return;
    }

    // ---- canonical codes + decoding table ---------------------------------------------------------------------------
    for (int i = lane; i < (1 << KNZ_HUF_MAXLEN) / 2; i += 64) ((uint32_t*)s_table)[i] = 0x00070007u;   // table[i] = 7 (:665-667)
    for (int s = lane; s < 256; s += 64) {
        const uint32_t ls = s_len[s];
        if (ls == 0) continue;
        uint32_t acc = 0, rank = 0;
        for (int u = 0; u < 256; u++) {
            const uint32_t lu = s_len[u];
            const bool before = lu != 0 && (lu < ls || (lu == ls && u < s));
            acc += before ? (1u << (KNZ_HUF_MAXLEN - lu)) : 0u;
            rank += before ? 1u : 0u;
        }
        s_C[rank] = (uint16_t)min(acc, 0xFFFFu);
        s_symAt[rank] = (uint8_t)s;
    }
    wave_sync();
    bool bad = false;
    for (int r = 0; r < count; r++) {
        const uint32_t s = s_symAt[r];
        const uint32_t len = s_len[s];
        const uint32_t size = 1u << (KNZ_HUF_MAXLEN - len);
        const uint32_t C = s_C[r];
        if (C + size > (1u << KNZ_HUF_MAXLEN)) { bad = true; break; }      // buildDecodingTable returns false (:683-685)
        const uint16_t val = (uint16_t)((s << 8) | len);
        for (uint32_t e = lane; e < size; e += 64) s_table[C + e] = val;
    }
    wave_sync();
    if (bad) { if (lane == 0) a.blk_status[b] = KNZ_ERR_PROCESS_BLOCK; return; }

    // ---- fragments: lane j decodes fragment j (:832-969) --------------------------------------------------------------
    const uint32_t F = n >> 2;
    if (lane < 4) {
        KnzStreamReader r;
        r.init(a.stream, a.nbytes, s_fragpos[lane]);
        uint8_t* d = dst + (size_t)lane * F;
        if ((((uintptr_t)d) & 3) == 0) {            // full chunks: fragments start 4096 bytes apart -> 32-bit stores
            uint32_t i = 0;
            for (; i + 4 <= F; i += 4) {
                uint32_t out = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t val = s_table[r.peek(KNZ_HUF_MAXLEN)];
                    r.consume(val & 0xFF);
                    out |= (val >> 8) << (8 * u);
                }
                *(uint32_t*)(d + i) = out;
            }
            for (; i < F; i++) { const uint32_t val = s_table[r.peek(KNZ_HUF_MAXLEN)]; r.consume(val & 0xFF); d[i] = (uint8_t)(val >> 8); }
        } else {
            for (uint32_t i = 0; i < F; i++) {
                const uint32_t val = s_table[r.peek(KNZ_HUF_MAXLEN)];
                r.consume(val & 0xFF);
                d[i] = (uint8_t)(val >> 8);
            }
        }
    }
    if (lane == 4) {
        for (uint32_t i = 4 * F; i < n; i++)
            dst[i] = (uint8_t)(knz_fetch32(a.stream, (int64_t)(s_tailpos + 8ull * (i - 4 * F)), (int64_t)(a.nbytes << 3)) >> 24);
    }
}

// -e NONE (NullEntropyCodec.go:43-62): chunk bytes become the units verbatim (4 x 4096 bytes in u1..u4)
__global__ __launch_bounds__(256) void knz_raw_units_kernel(HufEncArgs a) {
    const int tid = threadIdx.x;
    const uint32_t cpb = a.chunks_per_block;
    const uint32_t b = blockIdx.x / cpb, k = blockIdx.x % cpb;
    const uint32_t postLen = a.blk_len[b];
    uint32_t* ubits = a.unit_bits + (size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK;
    if (tid < KNZ_UNITS_PER_CHUNK)
        a.unit_src[(size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK + tid] = tid == 0 ? 0u : (uint32_t)(KNZ_U0_BYTES + (tid - 1) * KNZ_FRAG_BYTES);
    if ((uint64_t)k * KNZ_HUF_CHUNK >= postLen) { if (tid < KNZ_UNITS_PER_CHUNK) ubits[tid] = 0; return; }
    const uint32_t n = min((uint32_t)KNZ_HUF_CHUNK, postLen - k * KNZ_HUF_CHUNK);
    const uint8_t* src = a.data + a.blk_off[b] + (size_t)k * KNZ_HUF_CHUNK;
    uint8_t* slot = a.scratch + (size_t)blockIdx.x * KNZ_CHUNK_STRIDE;
    for (uint32_t i = tid; i < n; i += 256) slot[KNZ_U0_BYTES + (size_t)(i >> 12) * KNZ_FRAG_BYTES + (i & 4095)] = src[i];
    if (tid < KNZ_UNITS_PER_CHUNK) {
        uint32_t bits = 0;
        if (tid >= 1) { uint32_t lo = (uint32_t)(tid - 1) * 4096u; bits = n > lo ? 8u * min(4096u, n - lo) : 0u; }
        ubits[tid] = bits;
    }
}
