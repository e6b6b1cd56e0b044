// LZ / LZX forward, second form: the hash table of LZCodec.go is taken out of the parse.
//
// The reference's table holds, for every hash value, the last position inserted with it, and the parse inserts EVERY position it passes (the
// position it stands on :327-329, the lazy probes :362-378, all positions inside a match :517-553) except the ones it jumps over once 64
// probes in a row found nothing (`srcIdx = srcIdx1 + (srcInc >> 6)`, :356-358). Insert order is position order inside every slot (a match
// re-inserts its interior in increasing order), so the table entry a position p reads is
//     cand(p) = the largest q < p with hash(q) == hash(p) that is not one of the jumped-over positions ("holes") still unfilled at that time.
// Without holes that is a function of the data alone: a stable sort of all positions of all blocks by (block, hash) puts q right in front of
// p. Three data-parallel kernels produce cand[] and cp8[] = the length of the common prefix of p and cand(p) (capped at 255) for every
// position; the parse kernel below then walks the block reading those two arrays and the source sequentially (scalar cache), with no
// dependent table access and no table maintenance. Holes are kept exact: the parse marks jumped-over positions in a bitmap (a coarse copy in
// LDS says whether a region has any), a candidate that is a hole is replaced by the next older position with the same hash (cand[cand[p]]
// ...), a match that reaches back over holes fills them (the reference inserts them then). Everything else is the parse of lz.hip, which
// stays as KNZ_LZ_CHAIN=1 and as the cross-check of this form.

struct LzPreArgs {
    uint32_t nblocks;
    const uint64_t* in_ptr; const uint32_t* in_len; const uint8_t* active;
    const uint32_t* gstart;        // [nblocks + 1] first global index of the block's hashed positions (plen = count - 16 of blocks that take part)
    uint32_t hash_log;
    uint32_t* keys; uint32_t* vals;
    uint32_t* cand; uint8_t* cp8;
};

__device__ __forceinline__ uint32_t knz_lz_hash(uint64_t v, unsigned rshift) { return (uint32_t)(((v << 24) * (uint64_t)0x1E35A7BD) >> rshift); }

// keys[g] = block << hash_log | hash(position), vals[g] = g ; grid (ceil(maxlen / 256), nblocks)
__global__ __launch_bounds__(256) void knz_lz_keys_kernel(LzPreArgs a) {
    const uint32_t b = blockIdx.y;
    const uint32_t g0 = a.gstart[b], plen = a.gstart[b + 1] - g0;
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= plen) return;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    a.keys[g0 + p] = (b << a.hash_log) | knz_lz_hash(knz_vle64(src + p), 64 - a.hash_log);
    a.vals[g0 + p] = g0 + p;
}
// behind the stable sort by key: the element in front of g in its (block, hash) group is its candidate
// (Round 4: the common prefixes are a second kernel in POSITION order. Computed here, in hash order, every element cost four scattered
// accesses - the candidate and the prefix byte written, the text read at the position and at the candidate -: 9.7 ms for 212 MB. In position
// order the text at the position and both arrays are streams, and candidates of neighbouring positions are often neighbours themselves.)
__global__ __launch_bounds__(256) void knz_lz_cand_kernel(LzPreArgs a, const uint32_t* skeys, const uint32_t* svals, uint32_t total) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint32_t key = skeys[i], g = svals[i], b = key >> a.hash_log;
    uint32_t q = 0;
    if (i > 0 && skeys[i - 1] == key) q = svals[i - 1] - a.gstart[b];
    a.cand[g] = q;
}
// cp8[g] = length of the common prefix of the position and its candidate, capped at 255 and at the end of the block; grid (ceil(maxlen / 256), nblocks)
__global__ __launch_bounds__(256) void knz_lz_cp_kernel(LzPreArgs a) {
    const uint32_t b = blockIdx.y;
    const uint32_t g0 = a.gstart[b], plen = a.gstart[b + 1] - g0;
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    if (p >= plen) return;
    const uint32_t q = a.cand[g0 + p];
    uint32_t cp = 0;
    if (q) {
        const uint8_t* src = (const uint8_t*)a.in_ptr[b];
        const uint32_t count = a.in_len[b];
        const uint32_t lim = min(255u, count - p);                                // (q < p: the window of q ends first... no: it starts earlier, so p bounds both)
        while (cp + 8 <= lim) {
            const uint64_t d = knz_vle64(src + p + cp) ^ knz_vle64(src + q + cp);
            if (d) { cp += (uint32_t)(__ffsll((unsigned long long)d) - 1) >> 3; break; }
            cp += 8;
        }
        if (cp + 8 > lim) while (cp < lim && src[p + cp] == src[q + cp]) cp++;
    }
    a.cp8[g0 + p] = (uint8_t)cp;
}

struct LzParArgs {
    LzArgs a;
    const uint32_t* gstart; const uint32_t* cand; const uint8_t* cp8;
    uint32_t* holes;               // [nblocks * hole_stride] one bit per position, zeroed by the host
    uint64_t hole_stride;          // words per block
    unsigned long long* prof;      // [nblocks * 8] cycle counts per part of the parse (KNZ_LZ_PROF diagnostics) or null
};

__global__ __launch_bounds__(64) void knz_lz_forward_par_kernel(LzParArgs pa) {
    __shared__ uint32_t s_coarse[2048];                                  // one bit per 2^cs positions: the region holds (or held) holes
    const LzArgs& a = pa.a;
    const int lane = threadIdx.x;
    const bool writer = lane == 0;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const int count = (int)a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const uint32_t maxEnc = count <= 1024 ? (uint32_t)count + 16 : (uint32_t)count + (uint32_t)count / 64;   // MaxEncodedLen :935-941
    if (a.out_cap < maxEnc || count < KNZ_LZ_MIN_BLOCK) { if (writer) { a.ok[b] = 0; a.out_len[b] = 0; } return; }   // :256-263
    uint8_t* tkBuf = a.tk + (size_t)b * a.buf_stride;
    uint8_t* mBuf = a.mb + (size_t)b * a.buf_stride;
    uint8_t* mLenBuf = a.ml + (size_t)b * a.buf_stride;
    const int tkCap = count / 5 > 256 ? count / 5 : 256;
    const int srcEnd = count - 16 - 2;
    int maxDist = KNZ_LZ_MAX_DIST2;
    uint32_t flag = 1;
    if (srcEnd < 4 * KNZ_LZ_MAX_DIST1) { maxDist = KNZ_LZ_MAX_DIST1; flag = 0; }
    const uint32_t dt = a.blk_dt ? a.blk_dt[b] : 0u;
    if (dt == 9u /* DT_SMALL_ALPHABET */) { if (writer) { a.ok[b] = 0; a.out_len[b] = 0; } return; }
    const int minMatch = dt == 6u /* DT_DNA */ ? 6 : 4;
    flag |= ((minMatch - 2) & 7) << 1;
    if (writer) dst[12] = (uint8_t)flag;
    const uint8_t* cand8 = (const uint8_t*)(pa.cand + pa.gstart[b]);      // candidates / common prefixes of this block's positions
    const uint8_t* cp8 = pa.cp8 + pa.gstart[b];
    int32_t* holes = (int32_t*)(pa.holes + (size_t)b * pa.hole_stride);
    unsigned cs = 6;
    while (((uint32_t)count >> cs) >= 65536u) cs++;
    for (int i = lane; i < 2048; i += 64) s_coarse[i] = 0;
    wave_sync();
    bool anyHoles = false;
    int maxHole = -1;
    int srcIdx = 0, dstIdx = 13, anchor = 0, mLenIdx = 0, mIdx = 0, tkIdx = 0;
    int repd0 = count, repd1 = count, repdIdx = 0, srcInc = 0;
    int status = 1;                                                 // 1 ok, 0 skip, <0 error

    // candidate of position p as the reference's table would give it: cand[] unless that position is a hole that is still open
#define KNZ_LZP_CAND(P) ((int)wave_sload_u32(cand8 + 4 * (size_t)(P)))
#define KNZ_LZP_CP(P) ((int)((wave_sload_u32((const uint8_t*)((uintptr_t)(cp8 + (P)) & ~(uintptr_t)3)) >> (8 * ((uintptr_t)(cp8 + (P)) & 3))) & 0xFFu))
    auto is_hole = [&](int q) -> bool {
        if (!((s_coarse[(uint32_t)q >> (cs + 5)] >> (((uint32_t)q >> cs) & 31)) & 1u)) return false;
        return ((wave_bcast((uint32_t)knz_agent_load_i32(holes + (q >> 5)), 0)) >> (q & 31)) & 1u;   // (device-scope load: the bits are set by atomics at L2)
    };
    auto true_cand = [&](int raw) -> int {
        int q = raw;
        if (anyHoles) while (q > 0 && q <= maxHole && is_hole(q)) q = KNZ_LZP_CAND(q);
        return q;
    };
    // findMatchLZX(p, ref, maxMatch) from the common prefix cp (< 255): whole 8-byte steps only, the first difference ends it (:593-607)
    auto len_from_cp = [&](int cp, int maxMatch) -> int { const int whole = maxMatch & ~7; return cp < whole ? cp : whole; };

    // (cycle counts per part of the parse: a build with -DKNZ_LZ_PROFILE and KNZ_LZ_PROF=1 in the environment; the stamps cost ~12 % of this
    // instruction-bound loop, so they are compiled out otherwise)
#if defined(KNZ_LZ_PROFILE) && !defined(KNZ_HIP_EMU)
#define KNZ_LZP_NOW() (pa.prof ? (unsigned long long)__builtin_readcyclecounter() : 0ull)
#else
#define KNZ_LZP_NOW() 0ull
#endif
    unsigned long long tLoad = 0, tMiss = 0, tSearch = 0, tEmit = 0, nMiss = 0, nMatch = 0;
    while (srcIdx < srcEnd) {
        const unsigned long long c0 = KNZ_LZP_NOW();
        int bestLen = 0;
        const int srcIdx1 = srcIdx + 1;
        const int nextPos = srcIdx1 + (srcInc >> 6);
        const int maxMatch = min(srcEnd - srcIdx1, KNZ_LZ_MAX_MATCH);
        const int minRef = max(srcIdx - maxDist, 0);
        const int refA = srcIdx1 - (repdIdx ? repd1 : repd0), refB = srcIdx1 - (repdIdx ? repd0 : repd1);
        // everything this step reads, requested at once (one wait): the 8 bytes here, candidate and common prefix of this position, the 4 bytes
        // at the two repeat candidates (read unconditionally from a clamped address: a conditional load gets a wait of its own)
        const uint8_t* pp = src + srcIdx;
        const uint8_t* pa_ = src + max(refA, 0);
        const uint8_t* pb_ = src + max(refB, 0);
        const uint8_t* pc_ = cp8 + srcIdx;
        const uint8_t* y0 = (const uint8_t*)((uintptr_t)pp & ~(uintptr_t)3);
        const uint8_t* y2 = cand8 + 4 * (size_t)srcIdx;
        const uint8_t* y3 = (const uint8_t*)((uintptr_t)pc_ & ~(uintptr_t)3);
        const uint8_t* y4 = (const uint8_t*)((uintptr_t)pa_ & ~(uintptr_t)3);
        const uint8_t* y5 = (const uint8_t*)((uintptr_t)pb_ & ~(uintptr_t)3);
        uint64_t l0 = wave_sload_u64_async(y0);
        uint32_t l1 = wave_sload_u32_async(y0 + 8);
        uint32_t l2 = wave_sload_u32_async(y2);
        uint32_t l3 = wave_sload_u32_async(y3);
        uint64_t l4 = wave_sload_u64_async(y4);
        uint64_t l5 = wave_sload_u64_async(y5);
        WAVE_SLOAD_WAIT5A(l0, l2, l3, l4, l5, y0, y2, y3, y4, y5);
        l1 = wave_pin_sgpr(l1);                                             // (issued with the others, valid behind the same wait)
        const uint32_t shp = ((uint32_t)(uintptr_t)pp & 3u) * 8u;
        const uint64_t p = shp ? ((l0 >> shp) | ((uint64_t)l1 << (64 - shp))) : l0;
        const int raw0 = (int)l2, cp0 = (int)((l3 >> (8 * ((uint32_t)(uintptr_t)pc_ & 3u))) & 0xFFu);
        const uint32_t vA = (uint32_t)(l4 >> (((uint32_t)(uintptr_t)pa_ & 3u) * 8u)), vB = (uint32_t)(l5 >> (((uint32_t)(uintptr_t)pb_ & 3u) * 8u));
        const unsigned long long c1 = KNZ_LZP_NOW();
        tLoad += c1 - c0;
        const int ref0 = true_cand(raw0);
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
            if (ref > minRef) {
                const int mm = min(srcEnd - srcIdx, KNZ_LZ_MAX_MATCH);
                if (ref == raw0 && cp0 < 255) { if (cp0 >= 4) { bestLen = len_from_cp(cp0, mm); found = bestLen >= minMatch; } }
                else if ((uint32_t)p == knz_sle32(src + ref)) { bestLen = knz_lz_match_wave(src, srcIdx, ref, mm, lane); found = bestLen >= minMatch; }
            }
            if (!found) {
                if (nextPos > srcIdx1) {                                     // positions jumped over: never inserted by the reference (until a match covers them)
                    for (int q0 = srcIdx1; q0 < nextPos; q0 += 64) {
                        const int q = q0 + lane;
                        if (q < nextPos) {
                            atomicOr((unsigned int*)&holes[q >> 5], 1u << (q & 31));
                            atomicOr(&s_coarse[(uint32_t)q >> (cs + 5)], 1u << (((uint32_t)q >> cs) & 31));
                        }
                    }
                    wave_sync_lds();                                        // (no wait for the global atomics: the loads that test these bits follow them to the same L2 channel)
                    wave_order_lanes();
                    anyHoles = true;
                    maxHole = nextPos - 1;
                }
                srcIdx = nextPos;
                srcInc++;
                repdIdx = 0;
                tMiss += KNZ_LZP_NOW() - c1; nMiss++;
                continue;
            }
            if (ref != srcIdx - repd0 && ref != srcIdx - repd1) {      // checkNext (:362-398)
                {
                    const int raw1 = KNZ_LZP_CAND(srcIdx1), cp1 = KNZ_LZP_CP(srcIdx1);
                    const int ref1 = true_cand(raw1);
                    // taking the probe needs a match of at least bestLen there: a shorter common prefix settles it without touching memory
                    if (ref1 > minRef + 1 && !(ref1 == raw1 && cp1 < 255 && cp1 < bestLen) &&
                        knz_sle32(src + srcIdx1 + bestLen - 3) == knz_sle32(src + ref1 + bestLen - 3)) {
                        const int bestLen1 = (ref1 == raw1 && cp1 < 255) ? len_from_cp(cp1, maxMatch) : knz_lz_match_wave(src, srcIdx1, ref1, maxMatch, lane);
                        if (bestLen1 >= bestLen) { ref = ref1; bestLen = bestLen1; srcIdx = srcIdx1; }
                    }
                }
                if (a.extra) {
                    const int srcIdx2 = srcIdx1 + 1;
                    const int raw2 = KNZ_LZP_CAND(srcIdx2), cp2 = KNZ_LZP_CP(srcIdx2);
                    const int ref2 = true_cand(raw2);
                    const int mm2 = min(srcEnd - srcIdx2, KNZ_LZ_MAX_MATCH);
                    if (ref2 > minRef + 2 && !(ref2 == raw2 && cp2 < 255 && cp2 < bestLen) &&
                        knz_sle32(src + srcIdx2 + bestLen - 3) == knz_sle32(src + ref2 + bestLen - 3)) {
                        const int bestLen2 = (ref2 == raw2 && cp2 < 255) ? len_from_cp(cp2, mm2) : knz_lz_match_wave(src, srcIdx2, ref2, mm2, lane);
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
            else srcIdx++;
        }
        const unsigned long long c2 = KNZ_LZP_NOW();
        tSearch += c2 - c1; nMatch++;
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
        // the reference inserts every position of the match now (:517-553): holes under it are holes no longer
        if (anyHoles && srcIdx + 1 <= maxHole) {
            const int hi = min(anchor, maxHole + 1);
            for (int q0 = srcIdx + 1; q0 < hi; q0 += 64) { const int q = q0 + lane; if (q < hi) atomicAnd((unsigned int*)&holes[q >> 5], ~(1u << (q & 31))); }
            wave_order_lanes();
        }
        srcIdx = anchor;
        tEmit += KNZ_LZP_NOW() - c2;
    }
    if (pa.prof && writer) { unsigned long long* r = pa.prof + (size_t)b * 8; r[0] = tLoad; r[1] = tMiss; r[2] = tSearch; r[3] = tEmit; r[4] = nMiss; r[5] = nMatch; }
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
            wave_sync();
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
#undef KNZ_LZP_CAND
#undef KNZ_LZP_CP
}
