// LZ (LZXCodec, `extra` = false for LZ, true for LZX) of kanzi bitstream v6 on gfx950.
// Replaces LZXCodec.Forward / findMatchLZX / emitLengthLZ / hash and LZXCodec.inverseV6
// (v2/transform/LZCodec.go:249-591, 593-607, 193-214, 238-246, 621-778).
//
// The bytes the forward transform writes are defined by a greedy parse whose every decision feeds the next one (a one-entry hash
// table that every visited position overwrites, two repeat distances, a +1 (+2) lazy probe, backward extension, skip
// acceleration): one chain of ~230 K visited positions per MiB, one wave per block. What a lone wave pays for on this part is
// measured (profiles/r02_lone_wave_latencies.md): ~2.5 ns per instruction and 100-400 ns per DEPENDENT memory access; the
// kernel is built around the second number:
//   * the source is read-only for the kernel, so everything the chain reads of it (the 8 bytes at the position, the 4 bytes
//     at the two repeat candidates and at the hash candidate) comes through the scalar cache as wave-uniform s_load's: no
//     vector loads, no v_readfirstlane, several of them in flight at once;
//   * the hash table entry of the NEXT position (the one visited if this position finds nothing) is fetched while this
//     position is being decided, so a run of literals pays one dependent access per position (the candidate's bytes)
//     instead of three (position bytes -> table entry -> candidate bytes); an entry fetched ahead is corrected when the
//     current position hashes to the same slot;
//   * match lengths are measured by the whole wave, 512 bytes per round (8 bytes per lane, one ballot), and so are the
//     backward extension (64 bytes per round), the literal copies and the hashing of the positions inside a match
//     (positions only grow, so the sequential "last writer wins" is an atomicMax);
//   * every position the parse inserts is larger than all earlier ones, so a table write is an atomicMax (order-free, no fence on
//     the chain) and a table read a relaxed load, both at workgroup scope (the XCD's own L2: agent scope costs ~1 us per access); only the burst of inserts behind a match is
//     drained before the chain goes on.
// Parallelism across blocks only (the format gives no other); the table (256 KiB / 2 MiB per block) lives in HBM/L2.
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
    const uint8_t* blk_dt;      // [nblocks] ctx["dataType"] (text.hip numbering) or null: DNA = min match 6, small alphabet = skip (:298-311)
};

// unaligned little-endian loads composed of byte loads (wave-uniform addresses: one transaction each)
__device__ __forceinline__ uint32_t knz_le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ uint64_t knz_le64(const uint8_t* p) { return (uint64_t)knz_le32(p) | ((uint64_t)knz_le32(p + 4) << 32); }

// wave-uniform little-endian reads of the (read-only) source through the scalar cache: the aligned dwords around p, shifted
__device__ __forceinline__ uint64_t knz_sle64(const uint8_t* p) {
    const uintptr_t a = (uintptr_t)p & ~(uintptr_t)3;
    const uint32_t sh = ((uint32_t)(uintptr_t)p & 3u) * 8u;
    const uint32_t w0 = wave_sload_u32((const uint8_t*)a), w1 = wave_sload_u32((const uint8_t*)a + 4), w2 = wave_sload_u32((const uint8_t*)a + 8);
    const uint64_t lo = ((uint64_t)w1 << 32) | w0;
    return sh ? ((lo >> sh) | ((uint64_t)w2 << (64 - sh))) : lo;
}
__device__ __forceinline__ uint32_t knz_sle32(const uint8_t* p) {
    const uintptr_t a = (uintptr_t)p & ~(uintptr_t)3;
    const uint32_t sh = ((uint32_t)(uintptr_t)p & 3u) * 8u;
    const uint32_t w0 = wave_sload_u32((const uint8_t*)a), w1 = wave_sload_u32((const uint8_t*)a + 4);
    return (uint32_t)((((uint64_t)w1 << 32) | w0) >> sh);
}

// findMatchLZX (:593-607) by the whole wave: lane l compares the 8 bytes at offset base + 8 l, the first lane that differs (or
// whose 8 bytes no longer fit under maxMatch) ends the match; 512 bytes per round
__device__ __forceinline__ int knz_lz_match_wave(const uint8_t* src, int a, int b, int maxMatch, int lane) {
    for (int base = 0;; base += 512) {
        const int off = base + 8 * lane;
        const bool valid = off + 8 <= maxMatch;
        const uint64_t diff = valid ? (knz_vle64(src + a + off) ^ knz_vle64(src + b + off)) : 0;
        const uint64_t stop = wave_ballot(!valid || diff != 0);
        if (stop) {
            const uint32_t l = (uint32_t)(__ffsll((unsigned long long)stop) - 1);
            const uint32_t dlo = wave_readlane((uint32_t)diff, l), dhi = wave_readlane((uint32_t)(diff >> 32), l);
            const uint64_t d = ((uint64_t)dhi << 32) | dlo;
            return base + 8 * (int)l + (d ? (int)((__ffsll((unsigned long long)d) - 1) >> 3) : 0);   // (d == 0: the lane ran out of room)
        }
    }
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

// hash table entries: the table belongs to this workgroup alone, so workgroup-scope relaxed atomics (served by the XCD's L2: a load
// issued after a write of the same wave sees it, no fence)
__device__ __forceinline__ int knz_lz_tab_load(const int32_t* p) { return (int)wave_bcast((uint32_t)knz_wg_load_i32(p), 0); }   // (lane 0 is the lane that writes the table)

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
    const uint32_t dt = a.blk_dt ? a.blk_dt[b] : 0u;
    if (dt == 9u /* DT_SMALL_ALPHABET */) { if (writer) { a.ok[b] = 0; a.out_len[b] = 0; } return; }
    const int minMatch = dt == 6u /* DT_DNA */ ? 6 : 4;
    flag |= ((minMatch - 2) & 7) << 1;
    if (writer) dst[12] = (uint8_t)flag;
    int srcIdx = 0, dstIdx = 13, anchor = 0, mLenIdx = 0, mIdx = 0, tkIdx = 0;
    int repd0 = count, repd1 = count, repdIdx = 0, srcInc = 0;
    int status = 1;                                                 // 1 ok, 0 skip, <0 error
    // fetched ahead for position pfPos: its 8 bytes, its hash, and its table entry, which stays a pending per-lane load (pfLoad)
    // until the position is visited, unless the entry is known without asking memory (pfKnown >= 0: the slot was just written)
    int pfPos = -1, pfKnown = -1;
    uint32_t pfHash = 0, pfLoad = 0;
    uint64_t pfBytes = 0;
#define KNZ_LZ_HASHV(V) ((uint32_t)((((V) << 24) * (uint64_t)0x1E35A7BD) >> rshift))

    while (srcIdx < srcEnd) {
        int bestLen = 0;
        const bool ahead = pfPos == srcIdx;
        const uint64_t p = ahead ? pfBytes : knz_sle64(src + srcIdx);
        const uint32_t h0 = ahead ? pfHash : KNZ_LZ_HASHV(p);
        const int ref0 = ahead ? (pfKnown >= 0 ? pfKnown : (int)wave_bcast(pfLoad, 0)) : knz_lz_tab_load(hashes + h0);
        if (writer) knz_wg_max_i32(&hashes[h0], srcIdx);
        const int srcIdx1 = srcIdx + 1;
        // fetch ahead for the position this one leads to when it finds no match (:356-358): its bytes, its hash, and a request for
        // its table entry that is left in flight (issued before the candidate reads below, so that it overlaps them)
        const int nextPos = srcIdx1 + (srcInc >> 6);
        pfPos = -1;
        if (nextPos < srcEnd) {
            pfBytes = knz_sle64(src + nextPos);
            pfHash = KNZ_LZ_HASHV(pfBytes);
            pfKnown = pfHash == h0 ? srcIdx : -1;
            pfLoad = (uint32_t)knz_wg_load_i32(hashes + pfHash);
            pfPos = nextPos;
        }
        const int maxMatch = min(srcEnd - srcIdx1, KNZ_LZ_MAX_MATCH);
        const int minRef = max(srcIdx - maxDist, 0);
        // the 4 bytes at the two repeat candidates and at the hash candidate: three scalar loads in flight together (one wait)
        const int refA = srcIdx1 - (repdIdx ? repd1 : repd0), refB = srcIdx1 - (repdIdx ? repd0 : repd1);
        const uint32_t vA = refA > minRef ? knz_sle32(src + refA) : 0, vB = refB > minRef ? knz_sle32(src + refB) : 0;
        const uint32_t vC = ref0 > minRef ? knz_sle32(src + ref0) : 0;
        int ref = refA;
        if (ref > minRef && (uint32_t)(p >> 8) == vA) {
            bestLen = knz_lz_match_wave(src, srcIdx1, ref, maxMatch, lane);
        } else {
            ref = refB;
            if (ref > minRef && (uint32_t)(p >> 8) == vB) bestLen = knz_lz_match_wave(src, srcIdx1, ref, maxMatch, lane);
        }
        if (bestLen < minMatch) {
            ref = ref0;
            bool found = false;
            if (ref > minRef && (uint32_t)p == vC) {
                bestLen = knz_lz_match_wave(src, srcIdx, ref, min(srcEnd - srcIdx, KNZ_LZ_MAX_MATCH), lane);
                found = bestLen >= minMatch;
            }
            if (!found) {
                srcIdx = nextPos;
                srcInc++;
                repdIdx = 0;
                continue;
            }
            if (ref != srcIdx - repd0 && ref != srcIdx - repd1) {      // checkNext (:362-398)
                uint32_t h1;
                int ref1;
                if (pfPos == srcIdx1) { h1 = pfHash; ref1 = pfKnown >= 0 ? pfKnown : (int)wave_bcast(pfLoad, 0); }
                else { h1 = KNZ_LZ_HASHV(knz_sle64(src + srcIdx1)); ref1 = h1 == h0 ? srcIdx : knz_lz_tab_load(hashes + h1); }
                if (writer) knz_wg_max_i32(&hashes[h1], srcIdx1);
                if (ref1 > minRef + 1 && knz_sle32(src + srcIdx1 + bestLen - 3) == knz_sle32(src + ref1 + bestLen - 3)) {
                    const int bestLen1 = knz_lz_match_wave(src, srcIdx1, ref1, maxMatch, lane);
                    if (bestLen1 >= bestLen) { ref = ref1; bestLen = bestLen1; srcIdx = srcIdx1; }
                }
                if (a.extra) {
                    const int srcIdx2 = srcIdx1 + 1;
                    const uint32_t h2 = KNZ_LZ_HASHV(knz_sle64(src + srcIdx2));
                    const int ref2 = h2 == h1 ? srcIdx1 : (h2 == h0 ? srcIdx1 - 1 : knz_lz_tab_load(hashes + h2));
                    if (writer) knz_wg_max_i32(&hashes[h2], srcIdx2);
                    if (ref2 > minRef + 2 && knz_sle32(src + srcIdx2 + bestLen - 3) == knz_sle32(src + ref2 + bestLen - 3)) {
                        const int bestLen2 = knz_lz_match_wave(src, srcIdx2, ref2, min(srcEnd - srcIdx2, KNZ_LZ_MAX_MATCH), lane);
                        if (bestLen2 >= bestLen) { ref = ref2; bestLen = bestLen2; srcIdx = srcIdx2; }
                    }
                }
            }
            // extend backwards (:400-405): 64 bytes per round
            for (;;) {
                const int room = min(srcIdx - anchor, ref - minRef);
                const bool same = lane < room && src[srcIdx - 1 - lane] == src[ref - 1 - lane];
                const uint64_t stop = wave_ballot(!same);
                const int k = stop ? (int)(__ffsll((unsigned long long)stop) - 1) : 64;
                bestLen += k; ref -= k; srcIdx -= k;
                if (k < 64) break;
            }
            if (bestLen > KNZ_LZ_MAX_MATCH) {
                srcIdx += bestLen - KNZ_LZ_MAX_MATCH;
                ref += bestLen - KNZ_LZ_MAX_MATCH;
                bestLen = KNZ_LZ_MAX_MATCH;
            }
        } else {
            if ((uint8_t)p == (uint8_t)knz_sle32(src + ref - 1) && bestLen < KNZ_LZ_MAX_MATCH) { bestLen++; ref--; }
            else {
                srcIdx++;
                const uint32_t h1 = pfPos == srcIdx ? pfHash : KNZ_LZ_HASHV(knz_sle64(src + srcIdx));
                if (writer) knz_wg_max_i32(&hashes[h1], srcIdx);
            }
        }
        pfPos = -1;                                                  // a match: the chain restarts behind it
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
        // every position inside the match is hashed (:517-553); positions grow, so the sequential "last writer wins" is a max. Nothing
        // waits for them: the table loads that follow are issued behind them by the same wave and reach the same L2 channel in order.
        for (int pos = srcIdx + 1 + lane; pos < anchor; pos += 64) knz_wg_max_i32(&hashes[KNZ_LZ_HASHV(knz_vle64(src + pos))], pos);
        wave_order_lanes();
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
