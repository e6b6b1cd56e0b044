// Bit-granular stream assembly on the device: the gfx950 form of kanzi's ordered, unaligned emission.
//   * per block:  block-local stream = mode byte [skip byte] postTransformLength [checksum] | entropy units
//                 (v2/io/CompressedStream.go:870-914)
//   * per stream: header (:429-519) | for each block (lw-3):5 written:lw <bits> (:951-976) | end marker (:593-594)
// Every codec kernel leaves fixed-stride scratch slots holding up to 5 bit-string units per chunk plus their
// bit counts. knz_layout_blocks_kernel scans the counts inside each block, knz_layout_stream_kernel scans the
// blocks and writes the framing fields, knz_gather_kernel funnel-shifts every unit to its final bit offset
// (the device equivalent of DefaultOutputBitStream.WriteArray's shifted copy, DefaultOutputBitStream.go:150-172).
// HBM traffic: compressed bytes read once from scratch and written once to the stream.
#include "bits.h"

struct LayoutArgs {
    uint32_t nblocks;
    uint32_t chunks_per_block;
    const uint32_t* unit_bits;      // [nblocks*CPB*5]
    const uint32_t* blk_len;        // [nblocks] post-transform length
    const uint32_t* blk_src_len;    // [nblocks] original block length
    const uint8_t* blk_copy;        // [nblocks] copy block (<= 15 bytes or skipped by -s); null = derive from the length
    const uint8_t* blk_skip;        // [nblocks] ByteTransformSequence skip flags
    const uint64_t* blk_cksum;      // [nblocks] (when checksum_bits != 0)
    uint32_t checksum_bits;
    uint32_t n_transforms;          // ByteTransformSequence.Len()
    uint32_t chunk_size;            // bytes of post-transform data per chunk slot (16384 for Huffman/ANS0)
    uint32_t payload_only;          // 1: no block header bits (single EntropyEncoder object, knz_entropy_encode)
    // outputs
    uint64_t* chunk_rel;            // [nblocks*CPB] bit offset of the chunk inside the block-local stream
    uint64_t* blk_written;          // [nblocks] bits of the block-local stream
    uint32_t* blk_hdr;              // [nblocks*6] {hdrBits, mode, 4 BE words of header bits}
};

// Block header fields as encode() writes them (:866-887): returns the number of header bits and packs them,
// MSB first, into hdr[0..3] (up to 8+8+32+64 = 112 bits).
__device__ __forceinline__ uint32_t knz_block_header(bool copyBlock, uint32_t postLen, uint32_t skipFlags,
                                                     uint32_t ntransforms, uint32_t cksumBits, uint64_t cksum,
                                                     uint32_t* hdr, uint32_t* modeOut) {
    uint32_t mode = 0;
    if (copyBlock) mode |= 0x80;                             // _SMALL_BLOCK_SIZE copy block (:773-776) or skipped by -s (:795-799)
    uint32_t dataSize = 1;
    if (postLen >= 256) dataSize = ((31u - (uint32_t)__builtin_clz(postLen)) >> 3) + 1;
    mode |= ((dataSize - 1) & 3) << 5;
    hdr[0] = hdr[1] = hdr[2] = hdr[3] = 0;
    KnzBitWriter bw;
    bw.init(hdr);
    if ((mode & 0x80) || ntransforms <= 4) {
        mode |= skipFlags >> 4;
        bw.put(mode, 8);
    } else {
        mode |= 0x10;
        bw.put(mode, 8);
        bw.put(skipFlags, 8);
    }
    bw.put(postLen, 8 * dataSize);
    if (cksumBits == 32) bw.put((uint32_t)cksum, 32);
    else if (cksumBits == 64) { bw.put((uint32_t)(cksum >> 32), 32); bw.put((uint32_t)cksum, 32); }
    *modeOut = mode;
    return bw.pos;
}

// one workgroup per block: exclusive scan of the chunk bit counts
__global__ __launch_bounds__(256) void knz_layout_blocks_kernel(LayoutArgs a) {
    __shared__ uint64_t s_wave[4];
    __shared__ uint64_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t cpb = a.chunks_per_block;
    const uint32_t postLen = a.blk_len[b];
    const uint32_t nchunks = (postLen + a.chunk_size - 1) / a.chunk_size;
    uint32_t hdr[4], mode;
    uint32_t hdrBits = knz_block_header(a.blk_copy ? a.blk_copy[b] != 0 : a.blk_src_len[b] <= 15, postLen, a.blk_skip[b], a.n_transforms, a.checksum_bits,
                                        a.checksum_bits ? a.blk_cksum[b] : 0, hdr, &mode);
    if (a.payload_only) hdrBits = 0;
    if (tid == 0) s_carry = hdrBits;
    __syncthreads();
    for (uint32_t base = 0; base < nchunks; base += 256) {
        const uint32_t k = base + tid;
        uint64_t bits = 0;
        if (k < nchunks) {
            const uint32_t* u = a.unit_bits + ((size_t)b * cpb + k) * KNZ_UNITS_PER_CHUNK;
            bits = (uint64_t)u[0] + u[1] + u[2] + u[3] + u[4];
        }
        // chunk bit counts are < 2^18, 256 of them < 2^26: scan in 32 bits, carry in 64
        const uint32_t incl = wave_scan_incl((uint32_t)bits);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint64_t off = s_carry;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        if (k < nchunks) a.chunk_rel[(size_t)b * cpb + k] = off + incl - bits;
        __syncthreads();
        if (tid == 255) s_carry = off + incl;
        __syncthreads();
    }
    if (tid == 0) {
        a.blk_written[b] = s_carry;
        uint32_t* h = a.blk_hdr + (size_t)b * 6;
        h[0] = hdrBits; h[1] = mode; h[2] = hdr[0]; h[3] = hdr[1]; h[4] = hdr[2]; h[5] = hdr[3];
    }
}

struct StreamArgs {
    uint32_t nblocks;
    uint32_t chunks_per_block;
    uint32_t chunk_size;
    const uint32_t* blk_len;
    const uint64_t* chunk_rel;
    const uint64_t* blk_written;
    const uint32_t* blk_hdr;        // [nblocks*6]
    uint32_t* dst_words;            // output stream viewed as BE words (4-byte aligned)
    uint64_t dst_cap_bits;
    uint64_t first_bit;             // where the first block's framing starts (after the stream header)
    uint32_t framed;                // 1: .knz framing (lw:5, written:lw) ; 0: per-block outputs at fixed byte stride
    uint64_t block_stride_bits;     // framed == 0: block b's local stream starts at b*stride
    uint32_t end_marker;            // 1: account for the 8-bit end marker
    uint32_t header_words[8];       // stream header bits (up to 208), BE words
    uint32_t header_bits;           // 0: no stream header
    // outputs
    uint64_t* blk_dst_bit;          // [nblocks] bit position of the block-local stream in dst
    uint64_t* total_bits;           // [2] {total bits incl. end marker, overflow flag}
    const int32_t* blk_status;
};

__device__ __forceinline__ void knz_or_bits(uint32_t* words, uint64_t bit, uint32_t value, uint32_t count) {
    // OR `count` (<=32) bits of value at stream bit position `bit`
    if (count == 0) return;
    if (count < 32) value &= (1u << count) - 1u;
    uint64_t w = bit >> 5;
    uint32_t off = (uint32_t)(bit & 31), room = 32 - off;
    // dst holds byte-swapped BE words
    if (count <= room) atomicOr(&words[w], knz_bswap32(value << (room - count)));
    else {
        uint32_t rem = count - room;
        atomicOr(&words[w], knz_bswap32(value >> rem));
        atomicOr(&words[w + 1], knz_bswap32(value << (32 - rem)));
    }
}

// Single workgroup: scan over blocks, zero every dst word that will be OR-ed (chunk/segment boundaries and
// framing fields), then OR the stream header, the per-block length fields and the block headers.
__global__ __launch_bounds__(256) void knz_layout_stream_kernel(StreamArgs a) {
    __shared__ uint64_t s_total;
    __shared__ uint64_t s_wsum[4];
    __shared__ uint64_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // scan over blocks: bit position of every block-local stream behind its (lw-3):5, written:lw framing (:951-959)
    if (tid == 0) s_carry = a.first_bit + (a.framed ? a.header_bits : 0);
    __syncthreads();
    for (uint32_t b0 = 0; b0 < a.nblocks; b0 += 256) {
        const uint32_t b = b0 + tid;
        uint64_t written = 0, sz = 0;
        uint32_t lw = 3;
        if (b < a.nblocks) {
            written = a.blk_written[b];
            if (written >= 8) lw = (31u - (uint32_t)__builtin_clz((uint32_t)(written >> 3))) + 4;
            sz = a.framed ? 5 + lw + written : 0;
        }
        // 64-bit inclusive scan across the workgroup (block-local streams are < 2^34 bits)
        uint64_t incl = sz;
        for (int d = 1; d < 64; d <<= 1) { const uint64_t t = wave_shfl64(incl, lane - d); if (lane >= d) incl += t; }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        uint64_t before = s_carry;
        for (int w = 0; w < wave; w++) before += s_wsum[w];
        if (b < a.nblocks) a.blk_dst_bit[b] = a.framed ? before + incl - written : (uint64_t)b * a.block_stride_bits;
        __syncthreads();
        if (tid == 255) s_carry = before + incl;
        __syncthreads();
    }
    if (tid == 0) {
        uint64_t pos = s_carry;
        if (!a.framed) pos = a.nblocks ? (uint64_t)(a.nblocks - 1) * a.block_stride_bits + a.blk_written[a.nblocks - 1] : 0;
        if (a.framed && a.end_marker) pos += 8;
        s_total = pos;
        a.total_bits[0] = pos;
        a.total_bits[1] = (pos > a.dst_cap_bits) ? 1 : 0;
    }
    __threadfence();
    __syncthreads();
    if (s_total > a.dst_cap_bits) return;     // caller reports ERR_WRITE_FILE; nothing is written
    const uint32_t cpb = a.chunks_per_block;
    // pass 1: zero boundary words
    if (a.framed) {
        const uint64_t hw0 = a.first_bit >> 5, hw1 = (a.first_bit + a.header_bits + 7) >> 5;
        for (uint64_t w = hw0 + tid; w <= hw1; w += 256) a.dst_words[w] = 0;
        if (tid == 0) { uint64_t e = (s_total - 1) >> 5; a.dst_words[e] = 0; if (e) a.dst_words[e - 1] = 0; }
    }
    for (uint32_t b = tid; b < a.nblocks; b += 256) {
        const uint64_t base = a.blk_dst_bit[b];
        const uint32_t hdrBits = a.blk_hdr[(size_t)b * 6];
        // framing field (<= 39 bits before base) and block header (<= 112 bits after base)
        const uint64_t lo = (base >= 64 ? base - 64 : 0) >> 5, hi = (base + hdrBits + 31) >> 5;
        for (uint64_t w = lo; w <= hi; w++) a.dst_words[w] = 0;
        const uint64_t e = base + a.blk_written[b];
        a.dst_words[e >> 5] = 0;
        if (e >= 32) a.dst_words[(e >> 5) - 1] = 0;
    }
    for (uint32_t idx = tid; idx < a.nblocks * cpb; idx += 256) {
        const uint32_t b = idx / cpb, k = idx % cpb;
        const uint32_t nchunks = (a.blk_len[b] + a.chunk_size - 1) / a.chunk_size;
        if (k >= nchunks) continue;
        const uint64_t p0 = a.blk_dst_bit[b] + a.chunk_rel[idx];
        a.dst_words[p0 >> 5] = 0;                         // first word of this chunk (shared with its predecessor)
        if (p0) a.dst_words[(p0 - 1) >> 5] = 0;           // last word of the predecessor when p0 is word aligned
    }
    __threadfence();
    __syncthreads();
    // pass 2: OR the fields
    if (a.framed && a.header_bits) {
        for (uint32_t i = tid; i * 32 < a.header_bits; i += 256) {
            uint32_t cnt = min(32u, a.header_bits - i * 32);
            knz_or_bits(a.dst_words, a.first_bit + (uint64_t)i * 32, a.header_words[i] >> (32 - cnt), cnt);
        }
    }
    for (uint32_t b = tid; b < a.nblocks; b += 256) {
        const uint64_t base = a.blk_dst_bit[b];
        const uint64_t written = a.blk_written[b];
        if (a.framed) {
            uint32_t lw = 3;
            if (written >= 8) lw = (31u - (uint32_t)__builtin_clz((uint32_t)(written >> 3))) + 4;
            uint64_t p = base - lw - 5;
            knz_or_bits(a.dst_words, p, lw - 3, 5);
            // written on lw (<= 34) bits
            if (lw > 32) { knz_or_bits(a.dst_words, p + 5, (uint32_t)(written >> 32), lw - 32); knz_or_bits(a.dst_words, p + 5 + (lw - 32), (uint32_t)written, 32); }
            else knz_or_bits(a.dst_words, p + 5, (uint32_t)written, lw);
        }
        const uint32_t* h = a.blk_hdr + (size_t)b * 6;
        uint32_t hdrBits = h[0];
        for (uint32_t i = 0; i * 32 < hdrBits; i++) {
            uint32_t cnt = min(32u, hdrBits - i * 32);
            knz_or_bits(a.dst_words, base + (uint64_t)i * 32, h[2 + i] >> (32 - cnt), cnt);
        }
    }
}

struct GatherArgs {
    uint32_t chunks_per_block;
    uint32_t chunk_size;
    const uint32_t* blk_len;
    const uint32_t* unit_bits;
    const uint8_t* scratch;
    uint32_t chunk_stride;          // bytes per chunk slot
    const uint32_t* unit_src;       // [slots*5] byte offset of each unit inside its slot
    const uint64_t* chunk_rel;
    const uint64_t* blk_dst_bit;
    uint32_t* dst_words;
    const uint64_t* total_bits;     // [1] != 0 => overflow, skip
};

// One workgroup per chunk slot: every thread owns destination words and ORs in the contribution of each of the
// chunk's (up to 5) units, so only the chunk's first and last words are shared with other workgroups.
__global__ __launch_bounds__(256) void knz_gather_kernel(GatherArgs a) {
    const int tid = threadIdx.x;
    const uint32_t cpb = a.chunks_per_block;
    const uint32_t b = blockIdx.x / cpb, k = blockIdx.x % cpb;
    if (a.total_bits[1] != 0) return;
    if ((uint64_t)k * a.chunk_size >= a.blk_len[b]) return;
    const uint32_t* ub = a.unit_bits + (size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK;
    const uint8_t* slot = a.scratch + (size_t)blockIdx.x * a.chunk_stride;
    const uint32_t* usrc = a.unit_src + (size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK;
    uint32_t uoff[KNZ_UNITS_PER_CHUNK];
#pragma unroll
    for (int j = 0; j < KNZ_UNITS_PER_CHUNK; j++) uoff[j] = usrc[j];
    uint64_t ustart[KNZ_UNITS_PER_CHUNK + 1];
    ustart[0] = a.blk_dst_bit[b] + a.chunk_rel[blockIdx.x];
#pragma unroll
    for (int j = 0; j < KNZ_UNITS_PER_CHUNK; j++) ustart[j + 1] = ustart[j] + ub[j];
    const uint64_t p0 = ustart[0], p1 = ustart[KNZ_UNITS_PER_CHUNK];
    if (p1 == p0) return;
    const uint64_t w0 = p0 >> 5, w1 = (p1 - 1) >> 5;
    // large chunks (rANS order 1: up to 5.8 MB) are split over gridDim.y workgroups
    const uint64_t span = (w1 - w0 + gridDim.y) / gridDim.y;
    const uint64_t wa = w0 + (uint64_t)blockIdx.y * span, wb = min(w1, wa + span - 1);
    for (uint64_t w = wa + tid; w <= wb; w += 256) {
        const int64_t wbit = (int64_t)(w << 5);
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < KNZ_UNITS_PER_CHUNK; j++) {
            const int64_t us = (int64_t)ustart[j], ue = (int64_t)ustart[j + 1];
            if (ue <= wbit || us >= wbit + 32 || ue == us) continue;
            v |= knz_fetch32_unit(slot, uoff[j], wbit - us, ue - us);
        }
        const uint32_t sw = knz_bswap32(v);
        if (w == w0 || w == w1) atomicOr(&a.dst_words[w], sw);
        else a.dst_words[w] = sw;
    }
}
