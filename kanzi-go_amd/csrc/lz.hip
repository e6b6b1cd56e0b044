// LZ (LZXCodec, `extra` = false for LZ, true for LZX) of kanzi bitstream v6 on gfx950.
// Replaces LZXCodec.Forward / findMatchLZX / emitLengthLZ / hash and LZXCodec.inverseV6
// (v2/transform/LZCodec.go:249-591, 593-607, 193-214, 238-246, 621-778).
//
// The output of the forward transform is defined by the reference's SEQUENTIAL greedy parse (one-entry hash table that
// every visited position overwrites, two repeat distances, +1 (+2) lazy probe, backward extension, skip acceleration),
// so a block is one dependent chain: the kernel runs one wave per block, every lane executes the same (wave-uniform)
// parse so that the wave can help where the work is wide: literal copies, hashing of the positions inside a match
// (positions only grow, so "last writer wins" is an atomicMax) and, in the inverse, every match/literal copy
// (overlapping matches are periodic: byte i comes from ref + i % dist). Parallelism across blocks only; the hash table
// (256 KiB / 2 MiB per block) lives in HBM/L2. This is the slowest stage of the path on a GPU by construction and is
// measured as such.
#include "bits.h"

#define KNZ_LZ_MAX_DIST1 ((1 << 16) - 2)
#define KNZ_LZ_MAX_DIST2 ((1 << 24) - 2)
#define KNZ_LZ_MAX_MATCH (65535 + 254 + 4)
#define KNZ_LZ_MIN_BLOCK 24

struct LzArgs {
    uint32_t nblocks;
    const uint64_t* in_ptr; const uint32_t* in_len;
    const uint64_t* out_ptr; uint32_t out_cap;
    uint32_t* out_len; int32_t* ok; const uint8_t* active;
    int32_t* hashes;            // [nblocks << hashLog]
    uint8_t* tk; uint8_t* mb; uint8_t* ml;     // [nblocks * buf_stride] each
    uint64_t buf_stride;
    uint32_t extra;             // 1 = LZX
};

// unaligned little-endian loads composed of byte loads (wave-uniform addresses: one transaction each)
__device__ __forceinline__ uint32_t knz_le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ uint64_t knz_le64(const uint8_t* p) { return (uint64_t)knz_le32(p) | ((uint64_t)knz_le32(p + 4) << 32); }

__device__ __forceinline__ int knz_lz_find_match(const uint8_t* src, int srcIdx, int ref, int maxMatch) {   // :593-607
    int bestLen = 0;
    while (bestLen + 8 <= maxMatch) {
        const uint64_t diff = knz_le64(src + srcIdx + bestLen) ^ knz_le64(src + ref + bestLen);
        if (diff != 0) { bestLen += (int)(__ffsll((unsigned long long)diff) - 1) >> 3; break; }
        bestLen += 8;
    }
    return bestLen;
}

__device__ __forceinline__ int knz_lz_emit_length(uint8_t* block, int length, bool writer) {   // :193-214
    if (length < 254) { if (writer) block[0] = (uint8_t)length; return 1; }
    if (length < 65536 + 254) {
        length -= 254;
        if (writer) { block[0] = 254; block[1] = (uint8_t)(length >> 8); block[2] = (uint8_t)length; }
        return 3;
    }
    length -= 255;
    if (writer) { block[0] = 255; block[1] = (uint8_t)(length >> 16); block[2] = (uint8_t)(length >> 8); block[3] = (uint8_t)length; }
    return 4;
}

__global__ __launch_bounds__(64) void knz_lz_forward_kernel(LzArgs a) {
    const int lane = threadIdx.x;
    const bool writer = lane == 0;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const int count = (int)a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const uint32_t maxEnc = count <= 1024 ? (uint32_t)count + 16 : (uint32_t)count + (uint32_t)count / 64;   // MaxEncodedLen :935-941
    if (a.out_cap < maxEnc || count < KNZ_LZ_MIN_BLOCK) { if (writer) { a.ok[b] = 0; a.out_len[b] = 0; } return; }   // :256-263
    const unsigned hashLog = a.extra ? 19 : 16;
    const unsigned rshift = 64 - hashLog;
    int32_t* hashes = a.hashes + ((size_t)b << hashLog);
    uint8_t* tkBuf = a.tk + (size_t)b * a.buf_stride;
    uint8_t* mBuf = a.mb + (size_t)b * a.buf_stride;
    uint8_t* mLenBuf = a.ml + (size_t)b * a.buf_stride;
    const int tkCap = count / 5 > 256 ? count / 5 : 256;          // tkBuf is never grown by the reference (:275-285): overflow = Go panic
    const int srcEnd = count - 16 - 2;
    int maxDist = KNZ_LZ_MAX_DIST2;
    uint32_t flag = 1;
    if (srcEnd < 4 * KNZ_LZ_MAX_DIST1) { maxDist = KNZ_LZ_MAX_DIST1; flag = 0; }
    const int minMatch = 4;
    flag |= ((minMatch - 2) & 7) << 1;
    if (writer) dst[12] = (uint8_t)flag;
    int srcIdx = 0, dstIdx = 13, anchor = 0, mLenIdx = 0, mIdx = 0, tkIdx = 0;
    int repd0 = count, repd1 = count, repdIdx = 0, srcInc = 0;
    int status = 1;                                                 // 1 ok, 0 skip, <0 error
#define KNZ_LZ_HASH(P) ((uint32_t)(((knz_le64(P) << 24) * (uint64_t)0x1E35A7BD) >> rshift))

    while (srcIdx < srcEnd) {
        int bestLen = 0;
        const uint32_t h0 = KNZ_LZ_HASH(src + srcIdx);
        const int ref0 = hashes[h0];
        wave_sync();                                                // every lane has read the entry before lane 0 overwrites it
        if (writer) hashes[h0] = srcIdx;
        wave_sync();                                                // and sees the new value from here on
        const uint64_t p = knz_le64(src + srcIdx);
        const int srcIdx1 = srcIdx + 1;
        const int maxMatch = min(srcEnd - srcIdx1, KNZ_LZ_MAX_MATCH);
        int ref = srcIdx1 - (repdIdx ? repd1 : repd0);
        const int minRef = max(srcIdx - maxDist, 0);
        if (ref > minRef && (uint32_t)(p >> 8) == knz_le32(src + ref)) {
            bestLen = knz_lz_find_match(src, srcIdx1, ref, maxMatch);
        } else {
            ref = srcIdx1 - (repdIdx ? repd0 : repd1);
            if (ref > minRef && (uint32_t)(p >> 8) == knz_le32(src + ref)) bestLen = knz_lz_find_match(src, srcIdx1, ref, maxMatch);
        }
        if (bestLen < minMatch) {
            ref = ref0;
            bool found = false;
            if (ref > minRef && (uint32_t)p == knz_le32(src + ref)) {
                bestLen = knz_lz_find_match(src, srcIdx, ref, min(srcEnd - srcIdx, KNZ_LZ_MAX_MATCH));
                found = bestLen >= minMatch;
            }
            if (!found) {
                srcIdx = srcIdx1 + (srcInc >> 6);
                srcInc++;
                repdIdx = 0;
                continue;
            }
            if (ref != srcIdx - repd0 && ref != srcIdx - repd1) {      // checkNext (:362-398)
                const uint32_t h1 = KNZ_LZ_HASH(src + srcIdx1);
                const int ref1 = hashes[h1];
                wave_sync();
                if (writer) hashes[h1] = srcIdx1;
                wave_sync();
                if (ref1 > minRef + 1 && knz_le32(src + srcIdx1 + bestLen - 3) == knz_le32(src + ref1 + bestLen - 3)) {
                    const int bestLen1 = knz_lz_find_match(src, srcIdx1, ref1, maxMatch);
                    if (bestLen1 >= bestLen) { ref = ref1; bestLen = bestLen1; srcIdx = srcIdx1; }
                }
                if (a.extra) {
                    const int srcIdx2 = srcIdx1 + 1;
                    const uint32_t h2 = KNZ_LZ_HASH(src + srcIdx2);
                    const int ref2 = hashes[h2];
                    wave_sync();
                    if (writer) hashes[h2] = srcIdx2;
                    wave_sync();
                    if (ref2 > minRef + 2 && knz_le32(src + srcIdx2 + bestLen - 3) == knz_le32(src + ref2 + bestLen - 3)) {
                        const int bestLen2 = knz_lz_find_match(src, srcIdx2, ref2, min(srcEnd - srcIdx2, KNZ_LZ_MAX_MATCH));
                        if (bestLen2 >= bestLen) { ref = ref2; bestLen = bestLen2; srcIdx = srcIdx2; }
                    }
                }
            }
            while (srcIdx > anchor && ref > minRef && src[srcIdx - 1] == src[ref - 1]) { bestLen++; ref--; srcIdx--; }
            if (bestLen > KNZ_LZ_MAX_MATCH) {
                srcIdx += bestLen - KNZ_LZ_MAX_MATCH;
                ref += bestLen - KNZ_LZ_MAX_MATCH;
                bestLen = KNZ_LZ_MAX_MATCH;
            }
        } else {
            if (src[srcIdx] == src[ref - 1] && bestLen < KNZ_LZ_MAX_MATCH) { bestLen++; ref--; }
            else {
                srcIdx++;
                const uint32_t h1 = KNZ_LZ_HASH(src + srcIdx);
                wave_sync();
                if (writer) hashes[h1] = srcIdx;
                wave_sync();
            }
        }
        srcInc = 0;
        const int dist = srcIdx - ref;
        const int mLen = bestLen - minMatch;
        int token, mLenTh;
        if (dist == repd0) { token = 0x00; mLenTh = 3; }
        else if (dist == repd1) { token = 0x04; mLenTh = 3; }
        else {
            mLenTh = 7;
            if (dist >= 256) {
                if (dist >= 65536) { if (writer) { mBuf[mIdx] = (uint8_t)(dist >> 16); mBuf[mIdx + 1] = (uint8_t)(dist >> 8); } mIdx += 2; token = 0x18; }
                else { if (writer) mBuf[mIdx] = (uint8_t)(dist >> 8); mIdx++; token = 0x10; }
            } else token = 0x08;
            if (writer) mBuf[mIdx] = (uint8_t)dist;
            mIdx++;
        }
        if (mLen >= mLenTh) { token += mLenTh; mLenIdx += knz_lz_emit_length(mLenBuf + mLenIdx, mLen - mLenTh, writer); }
        else token += mLen;
        repd1 = repd0;
        repd0 = dist;
        repdIdx = 1;
        const int litLen = srcIdx - anchor;
        if (tkIdx >= tkCap) { status = -KNZ_ERR_PROCESS_BLOCK; break; }
        if (litLen == 0) { if (writer) tkBuf[tkIdx] = (uint8_t)token; tkIdx++; }
        else {
            if (litLen >= 7) {
                if (litLen >= (1 << 24)) { status = 0; break; }              // "too many literals" => skip
                if (writer) tkBuf[tkIdx] = (uint8_t)((7 << 5) | token);
                tkIdx++;
                dstIdx += knz_lz_emit_length(dst + dstIdx, litLen - 7, writer);
            } else { if (writer) tkBuf[tkIdx] = (uint8_t)((litLen << 5) | token); tkIdx++; }
            for (int i = lane; i < litLen; i += 64) dst[dstIdx + i] = src[anchor + i];      // emitLiteralsLZ, all lanes
            dstIdx += litLen;
        }
        anchor = srcIdx + bestLen;
        // every position inside the match is hashed (:517-553); positions grow, so the sequential "last writer wins" is a max
        wave_sync();
        for (int pos = srcIdx + 1 + lane; pos < anchor; pos += 64) atomicMax(&hashes[KNZ_LZ_HASH(src + pos)], pos);
        wave_sync();
        srcIdx = anchor;
    }
    if (status == 1) {
        const int litLen = count - anchor;
        if (dstIdx + litLen + tkIdx + mIdx >= count) status = 0;           // "no compression" (:559-561)
        else if (tkIdx >= tkCap) status = -KNZ_ERR_PROCESS_BLOCK;
        else {
            if (litLen >= 7) { if (writer) tkBuf[tkIdx] = (uint8_t)(7 << 5); tkIdx++; dstIdx += knz_lz_emit_length(dst + dstIdx, litLen - 7, writer); }
            else { if (writer) tkBuf[tkIdx] = (uint8_t)(litLen << 5); tkIdx++; }
            for (int i = lane; i < litLen; i += 64) dst[dstIdx + i] = src[anchor + i];
            dstIdx += litLen;
            if (writer) {
                const uint32_t v0 = (uint32_t)dstIdx, v1 = (uint32_t)tkIdx, v2 = (uint32_t)mIdx;
                for (int k = 0; k < 4; k++) { dst[k] = (uint8_t)(v0 >> (8 * k)); dst[4 + k] = (uint8_t)(v1 >> (8 * k)); dst[8 + k] = (uint8_t)(v2 >> (8 * k)); }
            }
            wave_sync();                                                    // lane 0's token/distance bytes are visible to the copy below
            __threadfence();
            for (int i = lane; i < tkIdx; i += 64) dst[dstIdx + i] = tkBuf[i];
            dstIdx += tkIdx;
            for (int i = lane; i < mIdx; i += 64) dst[dstIdx + i] = mBuf[i];
            dstIdx += mIdx;
            for (int i = lane; i < mLenIdx; i += 64) dst[dstIdx + i] = mLenBuf[i];
            dstIdx += mLenIdx;
            if (dstIdx > count - count / 100) status = 0;                 // :586-588
        }
    }
    if (writer) { a.ok[b] = status; a.out_len[b] = status == 1 ? (uint32_t)dstIdx : 0; }
#undef KNZ_LZ_HASH
}

__device__ __forceinline__ int knz_lz_read_length(const uint8_t* block, int& adv) {        // :216-232
    int res = block[0];
    if (res < 254) { adv = 1; return res; }
    if (res == 254) { res += (int)block[1] << 8; res += block[2]; adv = 3; return res; }
    res += (int)block[1] << 16; res += (int)block[2] << 8; res += block[3]; adv = 4; return res;
}

// A wave-uniform sequential reader: the next bytes of one of the block's streams staged in an LDS ring by the whole wave
// (coalesced loads, half a ring at a time), read back as LDS broadcasts. inverseV6 walks four such streams (literals with
// their inline length extensions, tokens, distances, match length extensions); reading them byte by byte from global memory
// put an L2 round trip on every step of the token chain.
template <int SIZE>
struct KnzLzRing {
    const uint8_t* src; uint8_t* ring; long long hi, limit;            // bytes [hi - SIZE, hi) are staged; reads past limit give 0
    __device__ __forceinline__ void init(const uint8_t* s, uint8_t* r, long long start, long long lim) {
        src = s; ring = r; limit = lim; hi = start & ~(long long)(SIZE / 2 - 1);
    }
    // makes bytes [idx, idx + need) available, need <= SIZE / 2
    __device__ __forceinline__ void ensure(long long idx, int need, int lane) {
        if (idx + need <= hi && idx >= hi - SIZE) return;
        if (idx >= hi + SIZE / 2 || idx < hi - SIZE) hi = idx & ~(long long)(SIZE / 2 - 1);   // far jump: restart the ring there
        while (idx + need > hi) {
            wave_sync_lds();
            for (int i = lane; i < SIZE / 2; i += 64) ring[(hi + i) & (SIZE - 1)] = (hi + i < limit) ? src[hi + i] : (uint8_t)0;
            hi += SIZE / 2;
            wave_sync_lds();
        }
    }
    __device__ __forceinline__ uint32_t at(long long idx) const { return ring[idx & (SIZE - 1)]; }
};

template <int SIZE>
__device__ __forceinline__ int knz_lz_read_length_ring(KnzLzRing<SIZE>& r, long long idx, int& adv, int lane) {   // :216-232
    r.ensure(idx, 4, lane);
    int res = (int)wave_uniform(r.at(idx));
    if (res < 254) { adv = 1; return res; }
    if (res == 254) { res += (int)wave_uniform(r.at(idx + 1)) << 8; res += (int)wave_uniform(r.at(idx + 2)); adv = 3; return res; }
    res += (int)wave_uniform(r.at(idx + 1)) << 16; res += (int)wave_uniform(r.at(idx + 2)) << 8; res += (int)wave_uniform(r.at(idx + 3)); adv = 4; return res;
}

// inverseV6 (:621-778): token driven; every copy is done by the whole wave
__global__ __launch_bounds__(64) void knz_lz_inverse_kernel(LzArgs a) {
    __shared__ uint8_t s_lit[4096], s_tk[1024], s_md[1024], s_ml[512];
    const int lane = threadIdx.x;
    const bool writer = lane == 0;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const long long count = (long long)a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    int status = 1;
    long long dstIdx = 0;
    if (count < 13) status = -KNZ_ERR_PROCESS_BLOCK;
    if (status == 1) {
        long long tkIdx = knz_le32(src), mIdx = knz_le32(src + 4), mLenIdx = knz_le32(src + 8);
        mIdx += tkIdx;
        mLenIdx += mIdx;
        if (tkIdx > count || mIdx > count || mLenIdx > count || tkIdx < 13) status = -KNZ_ERR_PROCESS_BLOCK;
        else {
            const long long srcEnd = tkIdx - 13;
            const int mFlag = src[12] & 1;
            const long long dstEnd = (long long)a.out_cap - 16;
            const long long maxDist = mFlag == 0 ? KNZ_LZ_MAX_DIST1 : KNZ_LZ_MAX_DIST2;
            const int minMatch = ((src[12] >> 1) & 7) + 2;
            long long srcIdx = 13;
            long long repd0 = count, repd1 = count;
            KnzLzRing<4096> lit; lit.init(src, s_lit, srcIdx, count);
            KnzLzRing<1024> tk; tk.init(src, s_tk, tkIdx, count);
            KnzLzRing<1024> md; md.init(src, s_md, mIdx, count);
            KnzLzRing<512> ml; ml.init(src, s_ml, mLenIdx, count);
            // output bytes below `visible` are known to be readable by every lane (a fence has passed since they were stored)
            long long visible = 0;
            for (;;) {
                if (tkIdx >= count) { status = -KNZ_ERR_PROCESS_BLOCK; break; }
                tk.ensure(tkIdx, 1, lane);
                const int token = (int)wave_uniform(tk.at(tkIdx));
                tkIdx++;
                if (token >= 32) {
                    long long litLen;
                    if (token >= 0xE0) {
                        if (srcIdx + 4 > count) { status = -KNZ_ERR_PROCESS_BLOCK; break; }
                        int adv; const int ll = knz_lz_read_length_ring(lit, srcIdx, adv, lane);
                        litLen = 7 + ll; srcIdx += adv;
                    } else litLen = token >> 5;
                    if (srcIdx + litLen > count || dstIdx + litLen > (long long)a.out_cap) { status = -KNZ_ERR_PROCESS_BLOCK; break; }
                    if (litLen <= 2048) {
                        lit.ensure(srcIdx, (int)litLen, lane);
                        for (long long i = lane; i < litLen; i += 64) dst[dstIdx + i] = (uint8_t)lit.at(srcIdx + i);
                    } else {
                        for (long long i = lane; i < litLen; i += 64) dst[dstIdx + i] = src[srcIdx + i];   // long run: straight copy
                    }
                    srcIdx += litLen;
                    dstIdx += litLen;
                    if (srcIdx >= srcEnd) break;
                }
                long long mLen, dist;
                const int f = token & 0x18;
                if (f == 0) {
                    mLen = token & 0x03;
                    if (mLen == 3) { if (mLenIdx + 4 > count + 3) { status = -KNZ_ERR_PROCESS_BLOCK; break; } int adv; const int mx = knz_lz_read_length_ring(ml, mLenIdx, adv, lane); mLen += minMatch + mx; mLenIdx += adv; }
                    else mLen += minMatch;
                    dist = (token & 0x04) == 0 ? repd0 : repd1;
                } else {
                    mLen = token & 0x07;
                    if (mLen == 7) { if (mLenIdx + 4 > count + 3) { status = -KNZ_ERR_PROCESS_BLOCK; break; } int adv; const int mx = knz_lz_read_length_ring(ml, mLenIdx, adv, lane); mLen += minMatch + mx; mLenIdx += adv; }
                    else mLen += minMatch;
                    if (mIdx + 3 > count + 2) { status = -KNZ_ERR_PROCESS_BLOCK; break; }
                    md.ensure(mIdx, 3, lane);
                    dist = (long long)wave_uniform(md.at(mIdx)); mIdx++;
                    if (f >= 0x10) {
                        dist = (dist << 8) | (long long)wave_uniform(md.at(mIdx)); mIdx++;
                        if (f == 0x18) { dist = (dist << 8) | (long long)wave_uniform(md.at(mIdx)); mIdx++; }
                    }
                }
                repd1 = repd0;
                repd0 = dist;
                const long long mEnd = dstIdx + mLen;
                const long long ref = dstIdx - dist;
                if (ref < 0 || dist > maxDist || mEnd > dstEnd || dist <= 0) { status = -KNZ_ERR_PROCESS_BLOCK; break; }
                // bytes stored since the last fence (by any lane) must be visible before they are read as match source; most
                // matches reach further back than that
                if (ref + (dist >= mLen ? mLen : dist) > visible) {
                    wave_sync();
                    __threadfence();
                    visible = dstIdx;
                }
                // overlapping match = periodic pattern of period dist: byte i comes from ref + i % dist (all already written)
                for (long long i = lane; i < mLen; i += 64) dst[dstIdx + i] = dst[ref + (dist >= mLen ? i : i % dist)];
                dstIdx = mEnd;
            }
            if (status == 1 && srcIdx != srcEnd + 13) status = -KNZ_ERR_PROCESS_BLOCK;
        }
    }
    if (writer) { a.ok[b] = status == 1 ? 1 : -KNZ_ERR_PROCESS_BLOCK; a.out_len[b] = status == 1 ? (uint32_t)dstIdx : 0; }
}
