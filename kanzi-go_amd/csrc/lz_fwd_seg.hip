// LZ / LZX forward, third form: the parse of lz_par.hip cut into segments that run side by side (LZXCodec.Forward,
// v2/transform/LZCodec.go:249-591).
//
// What the sequential parse carries from one position to the next is small: (srcIdx, anchor, repd[0], repd[1], repdIdx, srcInc)
// (:316-324) and which of the earlier positions were jumped over by the miss acceleration (:356-358) and never hashed ("holes";
// every other position's table entry is a function of the data: cand[] / cp8[] of lz_par.hip). It is also causal, and it forgets:
// behind a match srcInc is 0, repdIdx is 1, anchor = srcIdx, and two parses that emit the same two matches in a row agree on
// everything. So every segment of KNZ_LZS_SEG positions is parsed by a wave of its own from a GUESSED entry state, and the guesses
// are iterated to the fixed point "entry state of a segment = exit state of the one in front of it, holes read = holes written":
//   round:  parse the segments whose entry state is new, or that asked about holes which moved     [knz_lzs_parse_kernel]
//           keep the map stretches of the segments that did not run                                [knz_lzs_carry_kernel]
//           where did the hole maps change, and who had asked there?                               [knz_lzs_compare_kernel, knz_lzs_qhit_kernel]
//           new entry states: a sequential walk over the segments' (entry used, exit) pairs        [knz_lzs_relink_kernel]
// The first segment is exact in round 1, so segment k is exact after round k + 1 at the latest; in practice a trace started from a
// wrong state meets the true one after a match or two and a block settles in 3-5 rounds. A fixed point IS the sequential parse (by
// induction over the segments: exact entry state, exact holes in front of it), and only a fixed point is ever emitted; a block
// that has not settled after KNZ_LZS_MAX_ROUNDS goes to the one-wave kernel of lz_par.hip (counter KNZ_COUNTER_LZ_FWD_SERIAL_BLOCKS).
// Details that keep it exact:
//   * a parse leaves its segment only when it is not skipping (srcInc < 64): a long stretch without matches is walked by ONE wave
//     with the reference's growing stride (a 4 MiB stretch is ~23 K steps), the segments it runs over have nothing to do;
//   * holes are two set-only bit maps, J (jumped over) and M (inside a later match: hashed again, :517-553), hole = J & ~M, so
//     that no clear can race with a set; a segment reads its own generation for the positions it has passed itself and the
//     previous round's maps for everything in front of its entry anchor;
//   * tokens are written as descriptors (literal start, literal length, match length, distance, flag bits); literals, token bytes,
//     distances and length extensions are laid out afterwards by prefix sums over all tokens of the block [knz_lzs_emit_*].
#pragma once
#include "bits.h"

#define KNZ_LZS_SEG 4096u
#define KNZ_LZS_MAX_ROUNDS 48
#define KNZ_LZS_NEVER 0xFFFFFFFFu
#define KNZ_LZS_COARSE 512u                          // words of the coarse hole map (one bit per 2^cs positions)

struct LzSegArgs {
    LzParArgs pa;                  // source blocks, cand[], cp8[] (pa.holes unused here)
    uint32_t seg_size, segs;       // positions per segment, segments per block covered by the grid
    uint32_t tok_cap;              // token descriptors per segment
    uint32_t warm;                 // positions in front of its segment the first guess of a parse starts at
    uint32_t* entry;               // [nblocks][segs][5] srcIdx, anchor, repd0, repd1, srcInc | repdIdx << 31
    uint32_t* used;                // the entry state the segment's last parse started from (srcIdx = KNZ_LZS_NEVER: none yet)
    uint32_t* exit_;               // what that parse ended with
    uint32_t* ntok;                // [nblocks][segs]
    uint8_t* need;                 // [nblocks][segs] parse in this round
    uint4* tok;                    // [nblocks][segs][tok_cap] {literal start, literal length, match length | flag << 24, distance}
    uint32_t* Jp; uint32_t* Mp; uint32_t* Jn; uint32_t* Mn;     // hole maps of the previous / this round, [nblocks][map_stride]
    uint32_t* Cp; uint32_t* Cn;    // coarse maps [nblocks][KNZ_LZS_COARSE]: the region holds (or held) jumped-over positions
    uint32_t* Sp; uint32_t* Sn;    // [nblocks][2] any jumped-over position, the largest one
    uint64_t map_stride;
    uint8_t* blk_state;            // [nblocks] 0 running, 1 settled, 2 left to the one-wave kernel, 3 not a block for this stage, 4 declined before any parse (answered by the one-wave kernel)
    uint32_t* blk_flags;           // [nblocks][4] round results: entries changed, maps changed, rounds taken, first map word that moved
    uint32_t* rprof;               // diagnostics (KNZ_LZS_PROF) or null: [round][nblocks][3] first segment with a new entry state, first map word that moved, segments that run next
    uint32_t* qmap;                // [nblocks][segs][KNZ_LZS_COARSE] the coarse cells in front of its entry a segment's last parse asked about
    uint32_t* cmap;                // [nblocks][KNZ_LZS_COARSE] the coarse cells whose map words differ between the previous round and this one
    uint8_t* qhit;                 // [nblocks][segs] qmap & cmap != 0
    unsigned long long* sprof;     // diagnostics (KNZ_LZS_PROF) or null: [nblocks][segs][4] ticks of all parses, steps / hole-chain steps / ticks of the last one
    uint32_t all_again;            // every live segment of a block with holes runs in every round (the lane-per-segment parse; measurements of the other one)
    uint32_t max_rounds;           // a block that has not settled by then goes to the one-wave kernel
    // who has to run again when hole bits moved, found from the DATA (lane-per-segment parse; the wave-per-segment one keeps its query log): the words of the
    // maps that differ between two rounds [knz_lzs_compare_kernel], and for every bit of them the positions that could have asked about it = the next
    // positions with the same hash, as long as those are holes themselves [knz_lzs_mark_kernel]
    const uint32_t* skeys; const uint32_t* svals; uint32_t total, hash_log;   // the sorted (block << hash_log | hash, global position) pairs of lz_par.hip
    uint32_t* chg;                 // [nblocks][chg_cap] map words that moved in this round
    uint32_t* chg_n;               // [nblocks] their number (may exceed chg_cap: then every live segment of the block runs)
    uint32_t chg_cap;
};

__device__ __forceinline__ void knz_lzs_geom(const LzArgs& a, uint32_t b, int count, int& srcEnd, int& maxDist, int& minMatch, uint32_t& flag, bool& decline) {
    srcEnd = count - 16 - 2;
    maxDist = KNZ_LZ_MAX_DIST2; flag = 1;
    if (srcEnd < 4 * KNZ_LZ_MAX_DIST1) { maxDist = KNZ_LZ_MAX_DIST1; flag = 0; }
    const uint32_t dt = a.blk_dt ? a.blk_dt[b] : 0u;
    decline = dt == 9u;                                                   // DT_SMALL_ALPHABET (:306-308)
    minMatch = dt == 6u /* DT_DNA */ ? 6 : 4;
    flag |= ((minMatch - 2) & 7) << 1;
}

// a question about a position in front of the segment's entry: sets the cell's bit in the query log (.x) and returns the cell's bit of the coarse
// hole map (.y); every lane writes the same word and the wave is the only writer of its log, so no atomic is needed
__device__ __forceinline__ uint32_t knz_lzs_query_logged(uint32_t* qc, uint32_t q, uint32_t sh5, uint32_t sh) {
    const uint32_t cw = q >> sh5, cb = 1u << ((q >> sh) & 31);
    qc[2 * cw] |= cb;
    return wave_uniform(qc[2 * cw + 1] & cb);
}

// one thread per segment (grid (ceil(segs / 256), nblocks)): does the block take part, initial entry states
__global__ __launch_bounds__(256) void knz_lzs_init_kernel(LzSegArgs g) {
    const uint32_t b = blockIdx.y, s = blockIdx.x * 256 + threadIdx.x;
    const LzArgs& a = g.pa.a;
    if (b >= a.nblocks) return;
    if (s == 0) {
        g.blk_flags[4 * b] = g.blk_flags[4 * b + 1] = g.blk_flags[4 * b + 2] = 0; g.blk_flags[4 * b + 3] = 0xFFFFFFFFu;
        g.Sp[2 * b] = g.Sp[2 * b + 1] = g.Sn[2 * b] = g.Sn[2 * b + 1] = 0;
    }
    uint8_t st = 3;
    if (a.active[b]) {
        const int count = (int)a.in_len[b];
        const uint32_t maxEnc = count <= 1024 ? (uint32_t)count + 16 : (uint32_t)count + (uint32_t)count / 64;   // MaxEncodedLen :935-941
        int srcEnd, maxDist, minMatch; uint32_t flag; bool decline;
        knz_lzs_geom(a, b, count, srcEnd, maxDist, minMatch, flag, decline);
        // blocks the stage declines before it parses anything (:256-263, :306-308) are answered by the one-wave kernel
        st = (a.out_cap < maxEnc || count < KNZ_LZ_MIN_BLOCK || decline) ? 4 : ((((uintptr_t)a.in_ptr[b]) & 3) != 0 ? 2 : 0);
        if (st == 0 && s < g.segs) {
            const uint32_t ns = srcEnd > 0 ? ((uint32_t)srcEnd + g.seg_size - 1) / g.seg_size : 0;
            const size_t si = (size_t)b * g.segs + s;
            uint32_t* e = g.entry + 5 * si;
            // first guess: the state right behind a match that ended `warm` positions in front of the segment, repeat distances unknown: by
            // the time such a parse crosses into the segment it has usually met the true one (the positions it passes on the way belong to
            // the segment in front; what it writes there is replaced in the next round, when every segment gets its real entry state)
            const uint32_t at = s ? s * g.seg_size - min(g.warm, g.seg_size) : 0u;
            e[0] = at; e[1] = at; e[2] = (uint32_t)count; e[3] = (uint32_t)count; e[4] = s ? 0x80000000u : 0u;
            g.used[5 * si] = KNZ_LZS_NEVER;
            g.ntok[si] = 0;
            g.need[si] = s < ns ? 1 : 0;
        }
    }
    if (s == 0) g.blk_state[b] = st;
}

__global__ __launch_bounds__(64) void knz_lzs_parse_kernel(LzSegArgs g) {
    // per coarse word: .x = the cells in front of the entry state this parse asks about (its query log), .y = the coarse hole map; side by side, so
    // that a query is ONE LDS read (4 KiB: the waves of a CU are limited by their wave slots, not by LDS)
    __shared__ uint2 s_qc[KNZ_LZS_COARSE];
    const LzArgs& a = g.pa.a;
    const int lane = threadIdx.x;
    const bool writer = lane == 0;
    const uint32_t b = blockIdx.y, s = blockIdx.x;
    if (g.blk_state[b] != 0) return;
    const size_t si = (size_t)b * g.segs + s;
    if (!g.need[si]) return;
    const int count = (int)a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    int srcEnd, maxDist, minMatch; uint32_t hdrFlag; bool decline;
    knz_lzs_geom(a, b, count, srcEnd, maxDist, minMatch, hdrFlag, decline);
    const int segEnd = (int)min((uint64_t)(s + 1) * g.seg_size, (uint64_t)srcEnd);
    const uint32_t* E = g.entry + 5 * si;
    int srcIdx = (int)E[0], anchor = (int)E[1], repd0 = (int)E[2], repd1 = (int)E[3], srcInc = (int)(E[4] & 0x7FFFFFFFu), repdIdx = (int)(E[4] >> 31);
    if (writer) { uint32_t* U = g.used + 5 * si; U[0] = E[0]; U[1] = E[1]; U[2] = E[2]; U[3] = E[3]; U[4] = E[4]; }
    const int eSrc = srcIdx, eAnchor = anchor;                                // where this segment's own knowledge of the holes begins
    const uint8_t* cand8 = (const uint8_t*)(g.pa.cand + g.pa.gstart[b]);
    const uint8_t* cp8 = g.pa.cp8 + g.pa.gstart[b];
    const uint8_t* cpA = (const uint8_t*)((uintptr_t)cp8 & ~(uintptr_t)3);   // (4-byte aligned base + offset of the common-prefix bytes for the step's scalar read)
    const uint32_t cpo = (uint32_t)((uintptr_t)cp8 & 3);
    const uint32_t* Jp = g.Jp + (size_t)b * g.map_stride; const uint32_t* Mp = g.Mp + (size_t)b * g.map_stride;
    uint32_t* Jn = g.Jn + (size_t)b * g.map_stride; uint32_t* Mn = g.Mn + (size_t)b * g.map_stride;
    uint32_t* Cn = g.Cn + (size_t)b * KNZ_LZS_COARSE;
    unsigned cs = 6;
    while (((uint32_t)count >> cs) >= 32u * KNZ_LZS_COARSE) cs++;
    { const uint32_t* Cp = g.Cp + (size_t)b * KNZ_LZS_COARSE; for (int i = lane; i < (int)KNZ_LZS_COARSE; i += 64) { uint2 v; v.x = 0; v.y = Cp[i]; s_qc[i] = v; } }
    wave_sync();
    // What a parse takes from the previous round: whether the block had any jumped-over position (a change of that runs every segment
    // again), and the hole bits of the cells it asks about (logged in s_qc[].x; relink runs it again when one of those cells moved).
    // Nothing else: the largest hole only gates questions about positions this parse has passed itself.
    uint32_t anyHoles = g.Sp[2 * b] != 0 ? 1u : 0u;                          // (an integer, like the other wave-uniform flags the loop carries: a carried `bool` lives in a 64-bit lane mask and costs three scalar instructions wherever paths meet)
    int maxHole = -1;                                                         // the largest position this parse jumped over
    uint32_t ntok = 0;
    uint4* tokOut = g.tok + si * g.tok_cap;
    uint32_t overflow = 0;
#if !defined(KNZ_HIP_EMU)
#define KNZ_LZS_NOW() (unsigned long long)__builtin_readcyclecounter()
#else
#define KNZ_LZS_NOW() 0ull
#endif
    const unsigned long long t0 = g.sprof ? KNZ_LZS_NOW() : 0ull;
#ifdef KNZ_MEASURE
    uint32_t nSteps = 0, nChain = 0;                                          // (KNZ_LZS_PROF diagnostics: two scalar instructions per step that the product build does not pay)
#define KNZ_LZS_COUNT(X) (X)++
#else
    const uint32_t nSteps = 0, nChain = 0;
#define KNZ_LZS_COUNT(X) (void)0
#endif

// (an empty statement with side effects between two nested tests of wave-uniform values: without it the compiler merges them into one
// condition, which it evaluates as two 64-bit lane masks and an AND: eight scalar instructions where two compares and two branches do.
// Measured on S-silesia, 5 rounds of parse: 73.6 ms without any, 71.6 with the three in the repeat / candidate tests, 69.6 with the ones in the
// hole chain and the common-prefix test as well; more of them in the checkNext probes changed nothing. The loop-carried flags as integers
// instead of `bool` and one compare for the loop's exit were 77.2 -> 73.6 before that.)
#if !defined(KNZ_HIP_EMU)
#define KNZ_LZS_KEEP_NESTED() asm volatile("")
#else
#define KNZ_LZS_KEEP_NESTED() (void)0
#endif
#define KNZ_LZS_CAND(P) ((int)wave_sload_u32(cand8 + (size_t)(uint32_t)(4u * (uint32_t)(P))))                       // (32-bit offsets: the scalar load takes base + offset, no 64-bit address arithmetic)
#define KNZ_LZS_CP(P) ((int)((wave_sload_u32(cpA + (size_t)((cpo + (uint32_t)(P)) & ~3u)) >> (8 * ((cpo + (uint32_t)(P)) & 3u))) & 0xFFu))
    auto is_hole = [&](int q) -> bool {
        // (the form of these lines is measured, like the nesting statements: the log update and the coarse test of a foreign question as one helper, each
        // branch with its own wave-uniform result: 70.1 ms of parse; the same three steps written in line, 73.5; a hand-written 13-instruction block, 71.6)
        uint32_t hit;
        if (q < eSrc) hit = knz_lzs_query_logged((uint32_t*)s_qc, (uint32_t)q, cs + 5, cs);
        else {
            if (q > maxHole) return false;
            hit = wave_uniform(s_qc[(uint32_t)q >> (cs + 5)].y & (1u << (((uint32_t)q >> cs) & 31)));
        }
        if (hit == 0) return false;
        const uint32_t w = (uint32_t)q >> 5, m = 1u << (q & 31);
        uint32_t bits;
        if (q >= eAnchor) {                                                   // own generation (written by this wave: device-scope loads, the bits are set by atomics at L2)
            const uint32_t j = q >= eSrc ? wave_bcast((uint32_t)knz_agent_load_i32((const int32_t*)Jn + w), 0) : Jp[w];
            bits = j & ~wave_bcast((uint32_t)knz_agent_load_i32((const int32_t*)Mn + w), 0);
        } else bits = Jp[w] & ~Mp[w];
        return (bits & m) != 0;
    };
    auto true_cand = [&](int raw) -> int {
        int q = raw;
        if (anyHoles) while (q > 0) { KNZ_LZS_KEEP_NESTED(); if (!is_hole(q)) break; q = KNZ_LZS_CAND(q); KNZ_LZS_COUNT(nChain); }
        return q;
    };
    auto len_from_cp = [&](int cp, int maxMatch) -> int { const int whole = maxMatch & ~7; return cp < whole ? cp : whole; };

    // (Conditions of the hot path are written as nested single compares and integer flags: a combined or carried `bool` of wave-uniform values is
    // kept by the compiler as a 64-bit lane mask, four scalar instructions where a compare and a branch do, and the rounds are bound by the
    // scalar unit.)
    for (;;) {
        { const int stopAt = srcInc < 64 ? segEnd : srcEnd; if (srcIdx >= stopAt) break; }   // hand over only when not skipping (segEnd <= srcEnd; one select + one compare)
        KNZ_LZS_COUNT(nSteps);
        int bestLen = 0;
        const int srcIdx1 = srcIdx + 1;
        const int nextPos = srcIdx1 + (srcInc >> 6);
        const int maxMatch = min(srcEnd - srcIdx1, KNZ_LZ_MAX_MATCH);
        const int minRef = max(srcIdx - maxDist, 0);
        const int refA = srcIdx1 - (repdIdx ? repd1 : repd0), refB = srcIdx1 - (repdIdx ? repd0 : repd1);
        // the step's five reads as base + offset scalar loads (wave.h: wave_lz_step_loads)
        const uint32_t sIdx = (uint32_t)srcIdx, tcp = cpo + sIdx, ra = (uint32_t)max(refA, 0), rb = (uint32_t)max(refB, 0);
        const WaveLzLoads L = wave_lz_step_loads(src, cand8, cpA, sIdx & ~3u, sIdx << 2, tcp & ~3u, ra & ~3u, rb & ~3u);
        const uint64_t l0 = (uint64_t)L.a.x | ((uint64_t)L.a.y << 32);
        const uint32_t l1 = L.a.z;
        const uint32_t shp = (sIdx & 3u) * 8u;
        const uint64_t p = shp ? ((l0 >> shp) | ((uint64_t)l1 << (64 - shp))) : l0;
        const int raw0 = (int)L.c, cp0 = (int)((L.d >> (8 * (tcp & 3u))) & 0xFFu);
        const uint32_t vA = (uint32_t)(L.e >> ((ra & 3u) * 8u)), vB = (uint32_t)(L.f >> ((rb & 3u) * 8u));
        const int ref0 = true_cand(raw0);
        int ref = refB;                                                       // (what the reference leaves in `ref` when neither repeat distance matches)
        const uint32_t p1 = (uint32_t)(p >> 8);
        int rep = 0;                                                          // 1: the first repeat distance matches, 2: the second
        if (refA > minRef) { KNZ_LZS_KEEP_NESTED(); if (p1 == vA) rep = 1; }
        if (rep == 0) { if (refB > minRef) { KNZ_LZS_KEEP_NESTED(); if (p1 == vB) rep = 2; } }
        if (rep != 0) {
            ref = rep == 1 ? refA : refB;
            bestLen = knz_lz_match_wave(src, srcIdx1, ref, maxMatch, lane);
        }
        if (bestLen < minMatch) {
            ref = ref0;
            int found = 0;
            if (ref > minRef) {
                const int mm = min(srcEnd - srcIdx, KNZ_LZ_MAX_MATCH);
                int viaCp = 0;
                if (ref == raw0) { KNZ_LZS_KEEP_NESTED(); if (cp0 < 255) viaCp = 1; }
                if (viaCp) { KNZ_LZS_KEEP_NESTED(); if (cp0 >= 4) { bestLen = len_from_cp(cp0, mm); KNZ_LZS_KEEP_NESTED(); if (bestLen >= minMatch) found = 1; } }
                else if ((uint32_t)p == knz_sle32(src + ref)) { bestLen = knz_lz_match_wave(src, srcIdx, ref, mm, lane); if (bestLen >= minMatch) found = 1; }
            }
            if (found == 0) {
                if (nextPos > srcIdx1) {                                      // positions jumped over: not hashed until a match covers them
                    for (int q0 = srcIdx1; q0 < nextPos; q0 += 64) {
                        const int q = q0 + lane;
                        if (q < nextPos) {
                            atomicOr(&Jn[q >> 5], 1u << (q & 31));
                            atomicOr(&s_qc[(uint32_t)q >> (cs + 5)].y, 1u << (((uint32_t)q >> cs) & 31));
                            atomicOr(&Cn[(uint32_t)q >> (cs + 5)], 1u << (((uint32_t)q >> cs) & 31));
                        }
                    }
                    wave_sync_lds();
                    wave_order_lanes();
                    if (writer) { atomicOr(&g.Sn[2 * b], 1u); atomicMax(&g.Sn[2 * b + 1], (uint32_t)(nextPos - 1)); }
                    anyHoles = 1u;
                    maxHole = max(maxHole, nextPos - 1);
                }
                srcIdx = nextPos;
                srcInc++;
                repdIdx = 0;
                continue;
            }
            int checkNext = 0;                                             // checkNext (:362-398)
            if (ref != srcIdx - repd0) { if (ref != srcIdx - repd1) checkNext = 1; }
            if (checkNext) {
                {
                    const int raw1 = KNZ_LZS_CAND(srcIdx1), cp1 = KNZ_LZS_CP(srcIdx1);
                    const int ref1 = true_cand(raw1);
                    int probe = 0;
                    if (ref1 > minRef + 1) { if (!(ref1 == raw1 && cp1 < 255 && cp1 < bestLen)) probe = 1; }
                    if (probe) { if (knz_sle32(src + srcIdx1 + bestLen - 3) == knz_sle32(src + ref1 + bestLen - 3)) {
                        const int bestLen1 = (ref1 == raw1 && cp1 < 255) ? len_from_cp(cp1, maxMatch) : knz_lz_match_wave(src, srcIdx1, ref1, maxMatch, lane);
                        if (bestLen1 >= bestLen) { ref = ref1; bestLen = bestLen1; srcIdx = srcIdx1; }
                    } }
                }
                if (a.extra) {
                    const int srcIdx2 = srcIdx1 + 1;
                    const int raw2 = KNZ_LZS_CAND(srcIdx2), cp2 = KNZ_LZS_CP(srcIdx2);
                    const int ref2 = true_cand(raw2);
                    const int mm2 = min(srcEnd - srcIdx2, KNZ_LZ_MAX_MATCH);
                    if (ref2 > minRef + 2 && !(ref2 == raw2 && cp2 < 255 && cp2 < bestLen) &&
                        knz_sle32(src + srcIdx2 + bestLen - 3) == knz_sle32(src + ref2 + bestLen - 3)) {
                        const int bestLen2 = (ref2 == raw2 && cp2 < 255) ? len_from_cp(cp2, mm2) : knz_lz_match_wave(src, srcIdx2, ref2, mm2, lane);
                        if (bestLen2 >= bestLen) { ref = ref2; bestLen = bestLen2; srcIdx = srcIdx2; }
                    }
                }
            }
            for (;;) {                                                        // extend backwards (:400-405): 64 bytes per round
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
        srcInc = 0;
        const int dist = srcIdx - ref;
        uint32_t tflag;
        if (dist == repd0) tflag = 0x00;
        else if (dist == repd1) tflag = 0x04;
        else tflag = dist >= 65536 ? 0x18 : (dist >= 256 ? 0x10 : 0x08);
        repd1 = repd0;
        repd0 = dist;
        repdIdx = 1;
        if (ntok < g.tok_cap) { if (writer) { uint4 t; t.x = (uint32_t)anchor; t.y = (uint32_t)(srcIdx - anchor); t.z = (uint32_t)bestLen | (tflag << 24); t.w = (uint32_t)dist; tokOut[ntok] = t; } }
        else overflow = 1u;
        ntok++;
        anchor = srcIdx + bestLen;
        // the reference hashes every position of the match now (:517-553): jumped-over positions under it are holes no longer
        // (in a block with holes: all of them, so that the M bits of a trace do not depend on where other segments' holes were)
        if (anyHoles) {
            for (int q0 = srcIdx + 1; q0 < anchor; q0 += 64) { const int q = q0 + lane; if (q < anchor) atomicOr(&Mn[q >> 5], 1u << (q & 31)); }
            wave_order_lanes();
        }
        srcIdx = anchor;
    }
    if (writer) {
        uint32_t* X = g.exit_ + 5 * si;
        X[0] = (uint32_t)srcIdx; X[1] = (uint32_t)anchor; X[2] = (uint32_t)repd0; X[3] = (uint32_t)repd1; X[4] = (uint32_t)srcInc | ((uint32_t)repdIdx << 31);
        g.ntok[si] = overflow ? KNZ_LZS_NEVER : ntok;
    }
    if (g.sprof && writer) {
        const unsigned long long dt = KNZ_LZS_NOW() - t0;
        unsigned long long* P = g.sprof + 4 * si; P[0] += dt; P[1] = nSteps; P[2] = nChain; P[3] = dt;
    }
    wave_sync_lds();
    { uint32_t* Q = g.qmap + si * KNZ_LZS_COARSE; for (int i = lane; i < (int)KNZ_LZS_COARSE; i += 64) Q[i] = s_qc[i].x; }
#undef KNZ_LZS_CAND
#undef KNZ_LZS_KEEP_NESTED
#undef KNZ_LZS_COUNT
#undef KNZ_LZS_NOW
#undef KNZ_LZS_CP
}

// ---- the same parse, one LANE per segment (round 6) ---------------------------------------------------------------------------------------
// The wave-per-segment kernel above spends a whole wave on a chain of wave-uniform steps: ~125 instructions per step on the CU's ONE scalar unit,
// the 64-wide vector units idle (a full round over S-silesia = 156 M steps = 16-17 ms whatever the segment size). Here every lane walks a segment of
// its own: the step's arithmetic runs on the vector units 64 segments at a time; what a lane reads at its position and at the two repeat candidates
// sits in 16-byte register windows (KnzWin16) and, for cand[], in a 128-byte LDS line per lane; lanes that take different paths of the step serialise.
// Segments are short (KNZ_LZS_LANE_SEG positions) so that a wave's time - its slowest lane's chain - stays small and there are thousands of waves.
// Measured (S-silesia, 51 x 4 MiB): a full round 5.2-5.5 ms at 768 positions per segment against 16-17 ms for the wave-per-segment kernel (8-9 ms
// while the windows were picked apart with a computed register index, which the compiler turned into indexed scratch memory: KnzWin16::at).
// Who runs again: what the wave-per-segment form finds with its query log comes from the DATA here (knz_lzs_mark_kernel: the positions that can have
// asked about a hole bit that moved are the next positions with its hash), and a moved entry state makes its own segment and the one in front of it
// run, not the whole block (knz_lzs_walk_run, `guard`); the stretches of everybody else are carried. A block has settled when nobody has to run.
// S-silesia: 7 rounds, 5.6 + 5.3 + 2.9 + 2.4 + 2.1 + 1.9 + 0.6 = 20.7 ms (23 ms while the list of moved words held 8192 of them: a full third round); 62 ms (6 rounds) for the wave-per-segment form. 512 / 768 / 1024 / 2048 positions
// per segment: 21.5 / 22.6 / 28.1 / 39.9 ms of parse in 8 / 7 / 7 / 6 rounds (the other kernels of the stage grow with the number of segments).
// Same inputs, same outputs (entry / used / exit states, token descriptors, hole maps) and the same reads of the previous generation as
// knz_lzs_parse_kernel: the two are interchangeable (KNZ_LZS_WAVES selects the one above, with its query log).
#define KNZ_LZS_LANE_SEG 768u

__device__ __forceinline__ int knz_lz_match_lane(const uint8_t* src, int a, int b, int maxMatch) {     // findMatchLZX (:593-607): 8-byte compares, two per round trip
    int n = 0;
    while (n + 16 <= maxMatch) {
        const KnzPacked128* pa = (const KnzPacked128*)(src + a + n);
        const KnzPacked128* pb = (const KnzPacked128*)(src + b + n);
        const uint32_t a0 = pa->x, a1 = pa->y, a2 = pa->z, a3 = pa->w, b0 = pb->x, b1 = pb->y, b2 = pb->z, b3 = pb->w;
        const uint64_t d0 = ((uint64_t)(a1 ^ b1) << 32) | (a0 ^ b0), d1 = ((uint64_t)(a3 ^ b3) << 32) | (a2 ^ b2);
        if (d0) return n + (int)((__ffsll((unsigned long long)d0) - 1) >> 3);
        if (d1) return n + 8 + (int)((__ffsll((unsigned long long)d1) - 1) >> 3);
        n += 16;
    }
    while (n + 8 <= maxMatch) {
        const uint64_t d = knz_vle64(src + a + n) ^ knz_vle64(src + b + n);
        if (d) return n + (int)((__ffsll((unsigned long long)d) - 1) >> 3);
        n += 8;
    }
    return n;
}
// 16 bytes of a byte array kept in registers: a lane walks its segment position by position, and what it reads at the position (and at the position
// minus a repeat distance) moves with it. One 16-byte load serves the next 9-13 steps; a gather per step and array would fetch a whole cache line each
// time (the lanes of a wave sit a segment apart), and with 200 K lanes in flight those lines do not stay in any cache: the rounds were bound by HBM.
struct KnzWin16 {
    uint32_t w0, w1, w2, w3; int base;
    __device__ __forceinline__ void fill(const uint8_t* a, int pos) {
        base = pos & ~3;
        const KnzPacked128* q = (const KnzPacked128*)(a + base);
        w0 = q->x; w1 = q->y; w2 = q->z; w3 = q->w;
    }
    // the 64 bits that start `rel` bytes into the window (rel <= 8; rel <= 15 when fewer bits are used): shifts of the two halves, not a choice of
    // registers - a register picked by a computed index can make the compiler keep the window in scratch memory and index it there
    __device__ __forceinline__ uint64_t at(uint32_t rel) const {
        const uint64_t lo = (uint64_t)w0 | ((uint64_t)w1 << 32), hi = (uint64_t)w2 | ((uint64_t)w3 << 32);
        const uint32_t sh = rel * 8u;
        if (sh >= 64u) return hi >> (sh - 64u);
        return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
    }
    __device__ __forceinline__ uint32_t get32(const uint8_t* a, int pos) {          // the 4 bytes at pos (little endian)
        uint32_t rel = (uint32_t)(pos - base);
        if (rel > 12u) { fill(a, pos); rel = (uint32_t)(pos - base); }
        return (uint32_t)at(rel);
    }
    __device__ __forceinline__ uint64_t get64(const uint8_t* a, int pos) {          // the 8 bytes at pos
        uint32_t rel = (uint32_t)(pos - base);
        if (rel > 8u) { fill(a, pos); rel = (uint32_t)(pos - base); }
        return at(rel);
    }
    __device__ __forceinline__ uint32_t get8(const uint8_t* a, int pos) {
        uint32_t rel = (uint32_t)(pos - base);
        if (rel > 15u) { fill(a, pos); rel = (uint32_t)(pos - base); }
        return (uint32_t)at(rel) & 0xFFu;
    }
};

// bits [lo, hi) of a bit map set with one atomic per word
__device__ __forceinline__ void knz_lzs_set_bits(uint32_t* map, int lo, int hi) {
    for (int w = lo >> 5; w <= (hi - 1) >> 5; w++) {
        uint32_t m = 0xFFFFFFFFu;
        if (w == (lo >> 5)) m &= 0xFFFFFFFFu << (lo & 31);
        if (w == ((hi - 1) >> 5)) m &= 0xFFFFFFFFu >> (31 - ((hi - 1) & 31));
        atomicOr(&map[w], m);
    }
}

__global__ __launch_bounds__(64) void knz_lzs_parse_lanes_kernel(LzSegArgs g) {
    __shared__ uint32_t s_cp[KNZ_LZS_COARSE];                                  // the block's coarse hole map of the previous generation (all lanes of a workgroup walk segments of ONE block)
    const LzArgs& a = g.pa.a;
    const uint32_t b = blockIdx.y, s = blockIdx.x * 64 + threadIdx.x;
    if (g.blk_state[b] != 0) return;
    { const uint32_t* Cp = g.Cp + (size_t)b * KNZ_LZS_COARSE; for (uint32_t i = threadIdx.x; i < KNZ_LZS_COARSE; i += 64) s_cp[i] = Cp[i]; }
    wave_sync();
    if (s >= g.segs) return;
    const size_t si = (size_t)b * g.segs + s;
    if (!g.need[si]) return;
    const int count = (int)a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    int srcEnd, maxDist, minMatch; uint32_t hdrFlag; bool decline;
    knz_lzs_geom(a, b, count, srcEnd, maxDist, minMatch, hdrFlag, decline);
    const int segEnd = (int)min((uint64_t)(s + 1) * g.seg_size, (uint64_t)srcEnd);
    const uint32_t* E = g.entry + 5 * si;
    int srcIdx = (int)E[0], anchor = (int)E[1], repd0 = (int)E[2], repd1 = (int)E[3], srcInc = (int)(E[4] & 0x7FFFFFFFu), repdIdx = (int)(E[4] >> 31);
    { uint32_t* U = g.used + 5 * si; U[0] = E[0]; U[1] = E[1]; U[2] = E[2]; U[3] = E[3]; U[4] = E[4]; }
    const int eSrc = srcIdx, eAnchor = anchor;                                // where this segment's own knowledge of the holes begins
    const uint32_t* cand = g.pa.cand + g.pa.gstart[b];
    const uint8_t* cp8 = g.pa.cp8 + g.pa.gstart[b];
    const uint32_t* Jp = g.Jp + (size_t)b * g.map_stride; const uint32_t* Mp = g.Mp + (size_t)b * g.map_stride;
    uint32_t* Jn = g.Jn + (size_t)b * g.map_stride; uint32_t* Mn = g.Mn + (size_t)b * g.map_stride;
    uint32_t* Cn = g.Cn + (size_t)b * KNZ_LZS_COARSE;
    unsigned cs = 6;
    while (((uint32_t)count >> cs) >= 32u * KNZ_LZS_COARSE) cs++;
    bool anyHoles = g.Sp[2 * b] != 0;
    int maxHole = -1;                                                         // the largest position this parse jumped over
    // which stretches of its own trace hold a jumped-over position: one bit per seg_size / 32 positions behind the entry (everything further out in the
    // last bit). The trace's own hole bits live in the maps of this generation, which it sets with device-scope atomics and has to read back past the
    // caches (~1 us per read on this multi-XCD part): the mask answers "no hole there" for almost every question without asking memory.
    uint64_t ownMask = 0;
    unsigned os = 0;
    while ((g.seg_size >> os) > 32u) os++;
    uint32_t ntok = 0;
    uint4* tokOut = g.tok + si * g.tok_cap;
    bool overflow = false;
    KnzWin16 wS, wP, wA, wB;                                                  // source and cp8[] at the position, source at the two repeat candidates
    wS.base = wP.base = wA.base = wB.base = -0x40000000;
    const uint8_t* cand8 = (const uint8_t*)cand;
    // cand[] is the widest of the streams a lane walks (4 bytes per position): its current 128-byte line sits in LDS - fetched once, eight 16-byte loads
    // in flight, read from there for the next 32 positions (a 16-byte register window made every line come from HBM eight times: with it 175x the
    // algorithmic bytes and 23.7 ms of parse on S-silesia, with the line in LDS 131x and 22.5 ms; the source line in LDS as well: 100x, but 23.3 ms)
    __shared__ uint32_t s_cl[64 * 33];
    int clBase = -0x40000000;
    uint32_t* myCl = s_cl + 33 * threadIdx.x;
    auto cand_at = [&](int pos) -> int {
        uint32_t rel = (uint32_t)(pos - clBase);
        if (rel >= 32u) {
            const intptr_t a0 = (intptr_t)((uintptr_t)(cand8 + 4 * (size_t)pos) & ~(uintptr_t)127) - (intptr_t)cand8;      // the line the entry lies in
            clBase = (int)max((intptr_t)0, a0 >> 2);                          // (the workspace behind the last block's entries is readable: DevBuf keeps >= 256 bytes of slack)
            const KnzPacked128* q = (const KnzPacked128*)(cand8 + 4 * (size_t)clBase);
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const KnzPacked128 a = q[4 * half], b2 = q[4 * half + 1], c2 = q[4 * half + 2], d2 = q[4 * half + 3];
                uint32_t* o = myCl + 16 * half;
                o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b2.x; o[5] = b2.y; o[6] = b2.z; o[7] = b2.w;
                o[8] = c2.x; o[9] = c2.y; o[10] = c2.z; o[11] = c2.w; o[12] = d2.x; o[13] = d2.y; o[14] = d2.z; o[15] = d2.w;
            }
            rel = (uint32_t)(pos - clBase);
        }
        return (int)myCl[rel];
    };

    // the hole bit of position q as knz_lzs_parse_kernel reads it: the previous generation for everything in front of the entry anchor, J of the previous
    // generation with this trace's own M for the literal run it inherits, its own generation for what it has passed itself. (The coarse map only
    // filters: a cell without a jumped-over position in the previous generation has no hole to report.)
    auto is_hole = [&](int q) -> bool {
        const uint32_t w = (uint32_t)q >> 5, m = 1u << (q & 31);
        if (q >= eSrc) {
            if (q > maxHole) return false;
            if (((ownMask >> min((uint32_t)(q - eSrc) >> os, 63u)) & 1ull) == 0) return false;
            const uint32_t j = (uint32_t)knz_agent_load_i32((const int32_t*)Jn + w);
            if ((j & m) == 0) return false;
            return ((uint32_t)knz_agent_load_i32((const int32_t*)Mn + w) & m) == 0;
        }
        if ((s_cp[(uint32_t)q >> (cs + 5)] & (1u << (((uint32_t)q >> cs) & 31))) == 0) return false;
        if ((Jp[w] & m) == 0) return false;
        if (q >= eAnchor) return ((uint32_t)knz_agent_load_i32((const int32_t*)Mn + w) & m) == 0;
        return (Mp[w] & m) == 0;
    };
    auto true_cand = [&](int raw) -> int {
        int q = raw;
        if (anyHoles) while (q > 0 && is_hole(q)) q = (int)cand[q];
        return q;
    };
    auto len_from_cp = [&](int cp, int maxMatch) -> int { const int whole = maxMatch & ~7; return cp < whole ? cp : whole; };

    for (;;) {
        { const int stopAt = srcInc < 64 ? segEnd : srcEnd; if (srcIdx >= stopAt) break; }   // hand over only when not skipping
        int bestLen = 0;
        const int srcIdx1 = srcIdx + 1;
        const int nextPos = srcIdx1 + (srcInc >> 6);
        const int maxMatch = min(srcEnd - srcIdx1, KNZ_LZ_MAX_MATCH);
        const int minRef = max(srcIdx - maxDist, 0);
        const int refA = srcIdx1 - (repdIdx ? repd1 : repd0), refB = srcIdx1 - (repdIdx ? repd0 : repd1);
        const uint64_t p = wS.get64(src, srcIdx);
        const int raw0 = cand_at(srcIdx), cp0 = (int)wP.get8(cp8, srcIdx);
        const int ref0 = true_cand(raw0);
        int ref = refB;                                                       // (what the reference leaves in `ref` when neither repeat distance matches)
        const uint32_t p1 = (uint32_t)(p >> 8);
        int rep = 0;
        if (refA > minRef && p1 == wA.get32(src, refA)) rep = 1;
        else if (refB > minRef && p1 == wB.get32(src, refB)) rep = 2;
        if (rep != 0) {
            ref = rep == 1 ? refA : refB;
            bestLen = knz_lz_match_lane(src, srcIdx1, ref, maxMatch);
        }
        if (bestLen < minMatch) {
            ref = ref0;
            bool found = false;
            if (ref > minRef) {
                const int mm = min(srcEnd - srcIdx, KNZ_LZ_MAX_MATCH);
                if (ref == raw0 && cp0 < 255) { if (cp0 >= 4) { bestLen = len_from_cp(cp0, mm); found = bestLen >= minMatch; } }
                else if ((uint32_t)p == knz_vle32(src + ref)) { bestLen = knz_lz_match_lane(src, srcIdx, ref, mm); found = bestLen >= minMatch; }
            }
            if (!found) {
                if (nextPos > srcIdx1) {                                      // positions jumped over: not hashed until a match covers them
                    knz_lzs_set_bits(Jn, srcIdx1, nextPos);
                    for (uint32_t c = (uint32_t)srcIdx1 >> cs; c <= (uint32_t)(nextPos - 1) >> cs; c++) atomicOr(&Cn[c >> 5], 1u << (c & 31));
                    atomicOr(&g.Sn[2 * b], 1u); atomicMax(&g.Sn[2 * b + 1], (uint32_t)(nextPos - 1));
                    anyHoles = true;
                    maxHole = max(maxHole, nextPos - 1);
                    for (uint32_t c = min((uint32_t)(srcIdx1 - eSrc) >> os, 63u); c <= min((uint32_t)(nextPos - 1 - eSrc) >> os, 63u); c++) ownMask |= 1ull << c;
                }
                srcIdx = nextPos;
                srcInc++;
                repdIdx = 0;
                continue;
            }
            if (ref != srcIdx - repd0 && ref != srcIdx - repd1) {             // checkNext (:362-398)
                {
                    const int raw1 = cand_at(srcIdx1), cp1 = (int)wP.get8(cp8, srcIdx1);
                    const int ref1 = true_cand(raw1);
                    if (ref1 > minRef + 1 && !(ref1 == raw1 && cp1 < 255 && cp1 < bestLen) &&
                        knz_vle32(src + srcIdx1 + bestLen - 3) == knz_vle32(src + ref1 + bestLen - 3)) {
                        const int bestLen1 = (ref1 == raw1 && cp1 < 255) ? len_from_cp(cp1, maxMatch) : knz_lz_match_lane(src, srcIdx1, ref1, maxMatch);
                        if (bestLen1 >= bestLen) { ref = ref1; bestLen = bestLen1; srcIdx = srcIdx1; }
                    }
                }
                if (a.extra) {
                    const int srcIdx2 = srcIdx1 + 1;
                    const int raw2 = cand_at(srcIdx2), cp2 = (int)wP.get8(cp8, srcIdx2);
                    const int ref2 = true_cand(raw2);
                    const int mm2 = min(srcEnd - srcIdx2, KNZ_LZ_MAX_MATCH);
                    if (ref2 > minRef + 2 && !(ref2 == raw2 && cp2 < 255 && cp2 < bestLen) &&
                        knz_vle32(src + srcIdx2 + bestLen - 3) == knz_vle32(src + ref2 + bestLen - 3)) {
                        const int bestLen2 = (ref2 == raw2 && cp2 < 255) ? len_from_cp(cp2, mm2) : knz_lz_match_lane(src, srcIdx2, ref2, mm2);
                        if (bestLen2 >= bestLen) { ref = ref2; bestLen = bestLen2; srcIdx = srcIdx2; }
                    }
                }
            }
            {                                                                 // extend backwards (:400-405): 8 bytes per round trip
                int room = min(srcIdx - anchor, ref - minRef);
                while (room > 0 && ref >= 8) {
                    const uint64_t x = knz_vle64(src + srcIdx - 8) ^ knz_vle64(src + ref - 8);
                    const int m = min(room, x ? (int)(__clzll((long long)x) >> 3) : 8);
                    bestLen += m; ref -= m; srcIdx -= m; room -= m;
                    if (m < 8) { room = 0; break; }
                }
                while (room > 0 && src[srcIdx - 1] == src[ref - 1]) { bestLen++; ref--; srcIdx--; room--; }
            }
            if (bestLen > KNZ_LZ_MAX_MATCH) {
                srcIdx += bestLen - KNZ_LZ_MAX_MATCH;
                ref += bestLen - KNZ_LZ_MAX_MATCH;
                bestLen = KNZ_LZ_MAX_MATCH;
            }
        } else {
            if ((uint8_t)p == src[ref - 1] && bestLen < KNZ_LZ_MAX_MATCH) { bestLen++; ref--; }
            else srcIdx++;
        }
        srcInc = 0;
        const int dist = srcIdx - ref;
        uint32_t tflag;
        if (dist == repd0) tflag = 0x00;
        else if (dist == repd1) tflag = 0x04;
        else tflag = dist >= 65536 ? 0x18 : (dist >= 256 ? 0x10 : 0x08);
        repd1 = repd0;
        repd0 = dist;
        repdIdx = 1;
        if (ntok < g.tok_cap) { uint4 t; t.x = (uint32_t)anchor; t.y = (uint32_t)(srcIdx - anchor); t.z = (uint32_t)bestLen | (tflag << 24); t.w = (uint32_t)dist; tokOut[ntok] = t; }
        else overflow = true;
        ntok++;
        anchor = srcIdx + bestLen;
        // the reference hashes every position of the match now (:517-553): jumped-over positions under it are holes no longer
        if (anyHoles && anchor > srcIdx + 1) knz_lzs_set_bits(Mn, srcIdx + 1, anchor);
        srcIdx = anchor;
    }
    uint32_t* X = g.exit_ + 5 * si;
    X[0] = (uint32_t)srcIdx; X[1] = (uint32_t)anchor; X[2] = (uint32_t)repd0; X[3] = (uint32_t)repd1; X[4] = (uint32_t)srcInc | ((uint32_t)repdIdx << 31);
    g.ntok[si] = overflow ? KNZ_LZS_NEVER : ntok;
}

__device__ __forceinline__ bool knz_lzs_live(const LzSegArgs& g, uint32_t b, uint32_t s, int srcEnd) {
    const uint32_t segEnd = (uint32_t)min((uint64_t)(s + 1) * g.seg_size, (uint64_t)srcEnd);
    return (int)(s * (uint64_t)g.seg_size) < srcEnd && g.entry[5 * ((size_t)b * g.segs + s)] < segEnd;
}

// one wave per block: the entry states of the next round, and which segments run in it. The exit of a segment is known for the entry
// state its last parse started from; a segment whose entry state lies at or behind its end has nothing to parse and hands the state on
// unchanged.
//   * A block that has never had a jumped-over position has empty maps whatever its segments do: the segments whose entry state is new run.
//   * With holes, and an entry state that moved (or the block's first / last hole appearing): every live segment runs, so that the next
//     generation of the maps is the union of what ONE set of traces wrote, each over its own stretch of the block.
//   * With holes and no entry state moved, the traces are a chain (each starts where the one in front ended), the maps are exactly their
//     union, J over [entry srcIdx, exit srcIdx) and M over [entry anchor, exit anchor) per segment. Then only the segments that asked
//     about a cell whose words moved in this round run again (qhit); the others' stretches of the maps are copied into the next
//     generation [knz_lzs_carry_kernel]. A trace that is not run again is, by induction over the rounds, what a parse against the
//     latest maps would produce, so "nothing to run" is the same fixed point as before.
// (Round 4: one WAVE per block. The walk over the segments is sequential in the state it hands on, but nothing it reads depends on that state: the
// (used, exit) pairs of 512 segments are staged in LDS by all lanes, lane 0 walks them there, the new entry states go back in rows, and the
// "who runs" pass is one segment per lane. One thread per block did the same walk through ~15 dependent global reads per segment: 0.8 ms per round.)
// (Round 6: with one lane per segment a block has thousands of segments and lane 0's walk was 1.7 ms per round. The walk is sequential only in the
// state it hands on, and that state is almost always the exit of the segment in front. So every lane walks a run of consecutive segments on its
// own, from the exit state of the segment in front of its run; lane 0 then goes over the runs (256 of them) in order and walks again those whose real
// input turned out to be something else (a skipping trace that ran over the border of the run). A run walked from a wrong input may have
// forgotten traces (`used` reset) and raised `changed` for nothing: both only make segments run that did not have to.)
// `guard`: with the maps' dependents found from the data (g.chg), a moved entry state no longer makes every live segment of the block run. What it
// needs instead: where a segment's entry state moved (or a recorded trace died under a parse that ran over it), the maps of this round hold bits of TWO
// traces around that border - the stale one and the one in front of it. The stale one runs again because its entry is new; the one in front of it
// (the last live segment before it) is marked to run again as well, so that its stretch of the next generation is written afresh instead of being
// copied, stale bits and all, from this one. A stretch is carried only when both of its borders matched, i.e. nothing overlapped it.
struct LzsWalk { uint32_t cur[5]; bool changed, overflow; uint32_t lastLive; bool guardPrev; };
__device__ __forceinline__ void knz_lzs_walk_run(const LzSegArgs& g, uint32_t b, uint32_t s0, uint32_t s1, int srcEnd, LzsWalk& w) {
    for (uint32_t s = s0; s < s1; s++) {
        const size_t si = (size_t)b * g.segs + s;
        uint32_t* E = g.entry + 5 * si;
        uint32_t* U = g.used + 5 * si;
        const uint32_t* X = g.exit_ + 5 * si;
        const uint32_t u0 = U[0], u1 = U[1], u2 = U[2], u3 = U[3], u4 = U[4];
        for (int q = 0; q < 5; q++) E[q] = w.cur[q];
        const uint32_t segEnd = (uint32_t)min((uint64_t)(s + 1) * g.seg_size, (uint64_t)srcEnd);
        bool border = false;
        if (w.cur[0] >= segEnd) {                                                 // nothing to parse here: the state passes through
            // A trace this segment recorded while it still had something to parse is dead now, but its bits may sit in the maps: forget the trace and
            // have the parse that runs over it (and, without the list of moved words, every live segment) run, so that the next generation of the maps
            // is written by live traces only.
            if (u0 != KNZ_LZS_NEVER) { U[0] = KNZ_LZS_NEVER; w.changed = true; border = true; }
        } else {
            const bool same = u0 == w.cur[0] && u1 == w.cur[1] && u2 == w.cur[2] && u3 == w.cur[3] && u4 == w.cur[4];
            if (!same) { w.changed = true; border = true; }
            else if (g.ntok[si] == KNZ_LZS_NEVER) w.overflow = true;
            if (u0 != KNZ_LZS_NEVER) for (int q = 0; q < 5; q++) w.cur[q] = X[q]; // exact when `same`, the best guess otherwise
        }
        if (border && g.chg) {
            if (w.lastLive != KNZ_LZS_NEVER) g.qhit[(size_t)b * g.segs + w.lastLive] = 1;
            else w.guardPrev = true;                                              // (the last live segment lies in front of this run: lane 0 knows it)
        }
        if (E[0] < segEnd) w.lastLive = s;
    }
}
#define KNZ_LZS_RUNS 256u                                                          // runs a block's segments are cut into for the walk (= threads of the kernel)
__global__ __launch_bounds__(KNZ_LZS_RUNS) void knz_lzs_relink_kernel(LzSegArgs g, uint32_t round) {
    __shared__ uint32_t s_in[KNZ_LZS_RUNS * 5], s_out[KNZ_LZS_RUNS * 5];
    __shared__ uint32_t s_flag[KNZ_LZS_RUNS], s_last[KNZ_LZS_RUNS];
    __shared__ uint32_t s_res[8];
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const LzArgs& a = g.pa.a;
    if (b >= a.nblocks || g.blk_state[b] != 0) return;
    const int count = (int)a.in_len[b];
    const int srcEnd = count - 18;
    const uint32_t ns = srcEnd > 0 ? ((uint32_t)srcEnd + g.seg_size - 1) / g.seg_size : 0;
    const bool mapsChanged = g.blk_flags[4 * b + 1] != 0;
    const uint32_t per = (ns + KNZ_LZS_RUNS - 1) / KNZ_LZS_RUNS;                      // segments per run
    const uint32_t r0 = min(lane * per, ns), r1 = min(r0 + per, ns);
    {
        LzsWalk w;
        w.changed = false; w.overflow = false; w.lastLive = KNZ_LZS_NEVER; w.guardPrev = false;
        w.cur[0] = 0; w.cur[1] = 0; w.cur[2] = (uint32_t)count; w.cur[3] = (uint32_t)count; w.cur[4] = 0;
        if (r0 > 0 && r0 < ns) {                                                  // the guess: what the segment in front of the run left (read before any run is walked)
            const size_t sp = (size_t)b * g.segs + r0 - 1;
            if (g.used[5 * sp] != KNZ_LZS_NEVER) for (int q = 0; q < 5; q++) w.cur[q] = g.exit_[5 * sp + q];
            else w.cur[0] = KNZ_LZS_NEVER;                                        // (no guess: lane 0 walks this run with the real input)
        }
        for (int q = 0; q < 5; q++) s_in[5 * lane + q] = w.cur[q];
        __syncthreads();                                                          // (every guess is read before a walk forgets a trace)
        if (r0 < r1 && w.cur[0] != KNZ_LZS_NEVER) knz_lzs_walk_run(g, b, r0, r1, srcEnd, w);
        for (int q = 0; q < 5; q++) s_out[5 * lane + q] = w.cur[q];
        s_flag[lane] = (w.changed ? 1u : 0u) | (w.overflow ? 2u : 0u) | (w.guardPrev ? 4u : 0u);
        s_last[lane] = w.lastLive;
    }
    __threadfence();
    __syncthreads();
    if (lane == 0) {
        s_res[3] = 0;
        LzsWalk w;
        w.changed = false; w.overflow = false; w.lastLive = KNZ_LZS_NEVER; w.guardPrev = false;
        w.cur[0] = 0; w.cur[1] = 0; w.cur[2] = (uint32_t)count; w.cur[3] = (uint32_t)count; w.cur[4] = 0;
        for (uint32_t j = 0; j < KNZ_LZS_RUNS; j++) {
            const uint32_t q0 = min(j * per, ns), q1 = min(q0 + per, ns);
            if (q0 >= q1) break;
            const uint32_t* in = s_in + 5 * j;
            if (in[0] == w.cur[0] && in[1] == w.cur[1] && in[2] == w.cur[2] && in[3] == w.cur[3] && in[4] == w.cur[4]) {
                for (int q = 0; q < 5; q++) w.cur[q] = s_out[5 * j + q];
                w.changed = w.changed || (s_flag[j] & 1u) != 0; w.overflow = w.overflow || (s_flag[j] & 2u) != 0;
                if ((s_flag[j] & 4u) != 0 && w.lastLive != KNZ_LZS_NEVER) g.qhit[(size_t)b * g.segs + w.lastLive] = 1;
                if (s_last[j] != KNZ_LZS_NEVER) w.lastLive = s_last[j];
            } else {
                if (in[0] != KNZ_LZS_NEVER) {                                      // the run was walked from a state that was not its input: it may have forgotten traces (and its guards are not to be trusted)
                    w.changed = true;
                    if (g.chg) g.chg_n[b] = g.chg_cap + 1u;                        // (rare: everybody runs)
                }
                knz_lzs_walk_run(g, b, q0, q1, srcEnd, w);
            }
        }
        s_res[0] = w.changed ? 1u : 0u; s_res[1] = w.overflow ? 1u : 0u; s_res[2] = w.cur[1];
    }
    __threadfence();
    __syncthreads();
    const bool changed = s_res[0] != 0, overflow = s_res[1] != 0;
    const uint32_t lastAnchor = s_res[2];
    if (g.rprof && lane == 0) {
        uint32_t firstSeg = ns;
        for (uint32_t s = 0; s < ns; s++) {
            const size_t si = (size_t)b * g.segs + s;
            const uint32_t* E = g.entry + 5 * si;
            const uint32_t* U = g.used + 5 * si;
            const uint32_t segEnd = (uint32_t)min((uint64_t)(s + 1) * g.seg_size, (uint64_t)srcEnd);
            if (E[0] < segEnd && !(U[0] == E[0] && U[1] == E[1] && U[2] == E[2] && U[3] == E[3] && U[4] == E[4])) { firstSeg = s; break; }
        }
        uint32_t* R = g.rprof + ((size_t)round * a.nblocks + b) * 3;
        R[0] = firstSeg | (ns << 16); R[1] = g.blk_flags[4 * b + 3];
    }
    const bool holey = (g.Sp[2 * b] | g.Sn[2 * b]) != 0;
    const bool tooMany = g.chg != nullptr && g.chg_n[b] > g.chg_cap;                 // more map words moved than the list holds: nobody is singled out
    const bool everyone = holey && ((changed && (g.chg == nullptr || tooMany)) || (g.Sp[2 * b] != 0) != (g.Sn[2 * b] != 0) || ((g.all_again || tooMany) && mapsChanged));
    uint32_t nrun = 0;
    for (uint32_t s0 = 0; s0 < ns; s0 += KNZ_LZS_RUNS) {
        const uint32_t s = s0 + lane;
        bool run = false;
        if (s < ns) {
            const size_t si = (size_t)b * g.segs + s;
            const uint32_t segEnd = (uint32_t)min((uint64_t)(s + 1) * g.seg_size, (uint64_t)srcEnd);
            const uint32_t* E = g.entry + 5 * si;
            const uint32_t* U = g.used + 5 * si;
            const bool same = U[0] == E[0] && U[1] == E[1] && U[2] == E[2] && U[3] == E[3] && U[4] == E[4];
            run = E[0] < segEnd && (everyone || !same || (holey && !g.all_again && g.qhit[si] != 0));
            g.need[si] = run ? 1 : 0;
        }
        nrun += (uint32_t)__popcll((unsigned long long)wave_ballot(run));
    }
    if ((lane & 63) == 0 && nrun) atomicAdd(&s_res[3], nrun);
    __syncthreads();
    nrun = s_res[3];
    if (lane == 0) {
        if (g.rprof) g.rprof[((size_t)round * a.nblocks + b) * 3 + 2] = nrun;
        g.blk_flags[4 * b + 1] = 0;
        g.blk_flags[4 * b + 3] = 0xFFFFFFFFu;
        g.blk_flags[4 * b] = lastAnchor;                                          // anchor behind the last match (used when the block has settled)
        g.blk_flags[4 * b + 2] = round + 1;
        if (overflow) g.blk_state[b] = 2;
        else if (nrun == 0) g.blk_state[b] = 1;
        else if (round + 1 >= g.max_rounds) g.blk_state[b] = 2;
    }
}

// which map words differ from the previous round's, as coarse cells
// (Round 4: four words per thread through 16-byte reads; the cells and the "first word that moved" of a workgroup are combined in LDS and leave
// through a few global atomics; one atomic per moved word and two per wave on a handful of addresses per block were most of the kernel's 0.48 ms.)
// grid (ceil(words / 1024), nblocks); map_stride is a multiple of 4 words
__global__ __launch_bounds__(256) void knz_lzs_compare_kernel(LzSegArgs g, uint32_t words) {
    __shared__ uint32_t s_c[17];                                              // 1024 words = 32768 positions = at most 512 cells
    __shared__ uint32_t s_first, s_nchg, s_base;
    const uint32_t b = blockIdx.y;
    if (g.blk_state[b] != 0) return;
    if (threadIdx.x < 17) s_c[threadIdx.x] = 0;
    if (threadIdx.x == 17) { s_first = 0xFFFFFFFFu; s_nchg = 0; s_base = 0; }
    __syncthreads();
    const uint32_t w0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    uint32_t dmask = 0;
    if (w0 < words) {                                                         // (words is a multiple of 4 too)
        const size_t i = (size_t)b * g.map_stride + w0;
        const uint4 jp = *(const uint4*)(g.Jp + i), jn = *(const uint4*)(g.Jn + i), mp = *(const uint4*)(g.Mp + i), mn = *(const uint4*)(g.Mn + i);
        dmask = ((jp.x != jn.x || mp.x != mn.x) ? 1u : 0u) | ((jp.y != jn.y || mp.y != mn.y) ? 2u : 0u) |
                ((jp.z != jn.z || mp.z != mn.z) ? 4u : 0u) | ((jp.w != jn.w || mp.w != mn.w) ? 8u : 0u);
    }
    unsigned cs = 6;
    while ((g.pa.a.in_len[b] >> cs) >= 32u * KNZ_LZS_COARSE) cs++;
    const uint32_t cw0 = ((blockIdx.x * 1024u * 32u) >> cs) >> 5;            // first word of the coarse map this workgroup can touch
    uint32_t myOff = 0;
    if (dmask) {
        for (uint32_t k = 0; k < 4u; k++) if (dmask & (1u << k)) {
            const uint32_t cell = ((w0 + k) * 32u) >> cs;                    // a word of 32 positions lies inside one cell (cs >= 6)
            atomicOr(&s_c[(cell >> 5) - cw0], 1u << (cell & 31));
        }
        atomicMin(&s_first, w0 + (uint32_t)__ffs((int)dmask) - 1u);
        if (g.chg) myOff = atomicAdd(&s_nchg, (uint32_t)__popc(dmask));
    }
    __syncthreads();
    if (g.chg) {                                                             // the moved words of the workgroup go to the block's list in one reservation
        if (threadIdx.x == 0 && s_nchg) s_base = atomicAdd(&g.chg_n[b], s_nchg);
        __syncthreads();
        if (dmask) for (uint32_t k = 0; k < 4u; k++) if (dmask & (1u << k)) {
            const uint32_t at = s_base + myOff++;
            if (at < g.chg_cap) g.chg[(size_t)b * g.chg_cap + at] = w0 + k;
        }
    }
    __syncthreads();
    if (threadIdx.x < 17) { const uint32_t v = s_c[threadIdx.x]; if (v) atomicOr(&g.cmap[(size_t)b * KNZ_LZS_COARSE + cw0 + threadIdx.x], v); }
    if (threadIdx.x == 17 && s_first != 0xFFFFFFFFu) {
        g.blk_flags[4 * b + 1] = 1;
        atomicMin(&g.blk_flags[4 * b + 3], s_first);                          // (diagnostics: the first word that moved)
    }
}

// per segment: did its last parse ask about a cell that moved? grid (segs, nblocks), one wave
__global__ __launch_bounds__(64) void knz_lzs_qhit_kernel(LzSegArgs g) {
    const uint32_t b = blockIdx.y, s = blockIdx.x, lane = threadIdx.x;
    if (g.blk_state[b] != 0) return;
    const size_t si = (size_t)b * g.segs + s;
    const uint32_t* Q = g.qmap + si * KNZ_LZS_COARSE;
    const uint32_t* C = g.cmap + (size_t)b * KNZ_LZS_COARSE;
    uint32_t hit = 0;
    for (uint32_t i = lane; i < KNZ_LZS_COARSE; i += 64) hit |= Q[i] & C[i];
    const uint64_t any = wave_ballot(hit != 0);
    if (lane == 0) g.qhit[si] = any != 0 ? 1 : 0;
}

// Who asked about a hole bit that moved? A parse reads the bit of position q only on the way down a candidate chain: standing on p (or probing p + 1, p + 2
// from p) it looks at cand(p), and at cand(cand(p)) if that one is a hole, and so on. Seen from q: the NEXT position with q's hash asks about q, the one
// after it does if the next one is a hole, ... So for every bit that differs between the two generations the chain of successors is walked while it
// runs over holes (of either generation), and the traces that stand on a successor s (or on s - 1, s - 2: the probes) are marked to run again. The
// successor of a position comes from the sorted (block : hash, position) pairs the candidates were made from, by binary search.
__device__ __forceinline__ uint32_t knz_lzs_successor(const LzSegArgs& g, uint32_t b, const uint8_t* src, uint32_t g0, uint32_t plen, uint32_t q) {
    // the pairs are sorted by (block : hash, global position); the pairs of block b are [g0, g0 + plen)
    const uint32_t key = (b << g.hash_log) | knz_lz_hash(knz_vle64(src + q), 64 - g.hash_log);
    const uint64_t want = ((uint64_t)key << 32) | (g0 + q);
    uint32_t lo = g0, hi = g0 + plen;                                         // first pair >= want
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint64_t v = ((uint64_t)g.skeys[mid] << 32) | g.svals[mid];
        if (v < want) lo = mid + 1; else hi = mid;
    }
    if (lo + 1 >= g0 + plen || g.skeys[lo] != key || g.svals[lo] != g0 + q) return 0;
    return g.skeys[lo + 1] == key ? g.svals[lo + 1] - g0 : 0;
}
__device__ __forceinline__ void knz_lzs_mark_trace(const LzSegArgs& g, uint32_t b, uint32_t ns, uint32_t pos) {
    uint32_t t = min(pos / g.seg_size, ns - 1);
    for (;;) {                                                                // the trace that stands on pos: the last one that starts at or in front of it
        const uint32_t u0 = g.used[5 * ((size_t)b * g.segs + t)];
        if ((u0 != KNZ_LZS_NEVER && u0 <= pos) || t == 0) break;
        t--;
    }
    g.qhit[(size_t)b * g.segs + t] = 1;
}
// grid (up to 256, nblocks), 256 threads: one thread per POSITION of a moved word (a walk down the successors is a chain of binary searches,
// and a word has several positions that count: a thread per word would walk them one after the other); a workgroup takes 8 words at a time
__global__ __launch_bounds__(256) void knz_lzs_mark_kernel(LzSegArgs g) {
    const LzArgs& a = g.pa.a;
    const uint32_t b = blockIdx.y;
    if (g.blk_state[b] != 0) return;
    const uint32_t n = g.chg_n[b];
    if (n > g.chg_cap) return;                                                // (too many words moved: relink has every live segment run)
    const int count = (int)a.in_len[b];
    const int srcEnd = count - 18;
    const uint32_t ns = srcEnd > 0 ? ((uint32_t)srcEnd + g.seg_size - 1) / g.seg_size : 0;
    if (ns == 0) return;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    const uint32_t g0 = g.pa.gstart[b], plen = g.pa.gstart[b + 1] - g0;
    const size_t mi = (size_t)b * g.map_stride;
    for (uint32_t i = blockIdx.x * 8 + (threadIdx.x >> 5); i < n; i += gridDim.x * 8) {
        const uint32_t w = g.chg[(size_t)b * g.chg_cap + i];
        const uint32_t jp = g.Jp[mi + w], jn = g.Jn[mi + w], mp = g.Mp[mi + w], mn = g.Mn[mi + w];
        // a position counts when it is (or was) jumped over and one of its two bits moved: a parse reads J & ~M of the previous generation, or J alone of
        // the literal run it inherits (with its own M)
        const uint32_t rel = (jp | jn) & ((jp ^ jn) | (mp ^ mn));
        if (((rel >> (threadIdx.x & 31)) & 1u) == 0) continue;
        const uint32_t q = w * 32u + (threadIdx.x & 31);
        if (q >= plen) continue;
        uint32_t s = q;
        for (uint32_t hops = 0; hops < 4096u; hops++) {
            s = knz_lzs_successor(g, b, src, g0, plen, s);
            if (s == 0) break;
            knz_lzs_mark_trace(g, b, ns, s);
            if (s >= 1) knz_lzs_mark_trace(g, b, ns, s - 1);
            if (s >= 2) knz_lzs_mark_trace(g, b, ns, s - 2);
            const uint32_t sw = s >> 5, sm = 1u << (s & 31);
            if (((g.Jp[mi + sw] | g.Jn[mi + sw]) & sm) == 0) break;           // not a hole in either generation: nobody looks past it
            if (hops == 4095u) atomicAdd(&g.chg_n[b], g.chg_cap + 1u);        // (a chain of holes this long: give up on being selective, for the next round)
        }
    }
}

// the stretches of the maps that belong to live segments which did not run in this round, copied into this round's generation;
// grid (segs, nblocks), one wave, after the parse kernel
__global__ __launch_bounds__(64) void knz_lzs_carry_kernel(LzSegArgs g) {
    const LzArgs& a = g.pa.a;
    const uint32_t b = blockIdx.y, s = blockIdx.x, lane = threadIdx.x;
    if (g.blk_state[b] != 0) return;
    if ((g.Sp[2 * b]) == 0) return;                                           // nothing to copy
    const size_t si = (size_t)b * g.segs + s;
    const int count = (int)a.in_len[b];
    const int srcEnd = count - 18;
    if (g.need[si] || !knz_lzs_live(g, b, s, srcEnd)) return;
    const uint32_t* U = g.used + 5 * si;
    const uint32_t* X = g.exit_ + 5 * si;
    if (U[0] == KNZ_LZS_NEVER) return;
    unsigned cs = 6;
    while (((uint32_t)count >> cs) >= 32u * KNZ_LZS_COARSE) cs++;
    const uint32_t* Jp = g.Jp + (size_t)b * g.map_stride; const uint32_t* Mp = g.Mp + (size_t)b * g.map_stride;
    uint32_t* Jn = g.Jn + (size_t)b * g.map_stride; uint32_t* Mn = g.Mn + (size_t)b * g.map_stride;
    uint32_t* Cn = g.Cn + (size_t)b * KNZ_LZS_COARSE;
    uint32_t top = 0; bool any = false;
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t lo = pass ? U[1] : U[0], hi = pass ? X[1] : X[0];      // M over [entry anchor, exit anchor), J over [entry srcIdx, exit srcIdx)
        if (hi <= lo) continue;
        const uint32_t w0 = lo >> 5, w1 = (hi - 1) >> 5;
        for (uint32_t w = w0 + lane; w <= w1; w += 64) {
            uint32_t mask = 0xFFFFFFFFu;
            if (w == w0) mask &= 0xFFFFFFFFu << (lo & 31);
            if (w == w1) mask &= 0xFFFFFFFFu >> (31 - ((hi - 1) & 31));
            const uint32_t v = (pass ? Mp[w] : Jp[w]) & mask;
            if (v == 0) continue;
            atomicOr(pass ? &Mn[w] : &Jn[w], v);
            if (!pass) {
                const uint32_t cell = (w * 32u) >> cs;
                atomicOr(&Cn[cell >> 5], 1u << (cell & 31));
                any = true;
                top = max(top, w * 32u + 31u - (uint32_t)__clz(v));
            }
        }
    }
    if (any) { atomicOr(&g.Sn[2 * b], 1u); atomicMax(&g.Sn[2 * b + 1], top); }
}

// ---- layout of a settled block (:425-591) ------------------------------------------------------------------------------------
struct LzsSizes { uint32_t lit, dist, ml; bool special; };
__device__ __forceinline__ uint32_t knz_lzs_len_size(uint32_t length) { return length < 254u ? 1u : (length < 65536u + 254u ? 3u : 4u); }   // emitLengthLZ :193-212
__device__ __forceinline__ LzsSizes knz_lzs_sizes(const uint4& t, uint32_t minMatch) {
    LzsSizes z;
    const uint32_t litLen = t.y, bestLen = t.z & 0xFFFFFFu, fl = t.z >> 24;
    z.special = litLen >= (1u << 24);                                         // "too many literals" (:493-495): left to the one-wave kernel
    z.lit = litLen + (litLen >= 7 ? knz_lzs_len_size(litLen - 7) : 0u);
    z.dist = fl == 0x08 ? 1u : (fl == 0x10 ? 2u : (fl == 0x18 ? 3u : 0u));
    const uint32_t mLen = bestLen - minMatch, th = fl < 8 ? 3u : 7u;
    z.ml = mLen >= th ? knz_lzs_len_size(mLen - th) : 0u;
    return z;
}

// per segment: tokens, literal-stream bytes, distance bytes, length-extension bytes; grid (segs, nblocks)
__global__ __launch_bounds__(256) void knz_lzs_emit_count_kernel(LzSegArgs g, uint32_t* segsum) {
    __shared__ uint32_t s_w[4];
    const LzArgs& a = g.pa.a;
    const uint32_t b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    __shared__ uint32_t s_state;                                          // (other workgroups of the block may set the state to 2 meanwhile: read once per workgroup)
    if (tid == 0) s_state = g.blk_state[b];
    __syncthreads();
    if (s_state != 1) return;
    const int count = (int)a.in_len[b];
    int srcEnd, maxDist, minMatch; uint32_t hf; bool decline;
    knz_lzs_geom(a, b, count, srcEnd, maxDist, minMatch, hf, decline);
    const size_t si = (size_t)b * g.segs + s;
    const uint32_t n = knz_lzs_live(g, b, s, srcEnd) ? g.ntok[si] : 0u;
    const uint4* tk = g.tok + si * g.tok_cap;
    uint32_t lit = 0, dist = 0, ml = 0;
    bool special = false;
    for (uint32_t i = tid; i < n; i += 256) { const LzsSizes z = knz_lzs_sizes(tk[i], (uint32_t)minMatch); lit += z.lit; dist += z.dist; ml += z.ml; special |= z.special; }
    lit = knz_lzi_wg_sum(lit, s_w); dist = knz_lzi_wg_sum(dist, s_w); ml = knz_lzi_wg_sum(ml, s_w);
    if (special) g.blk_state[b] = 2;
    if (tid == 0) { uint32_t* S = segsum + 4 * si; S[0] = n; S[1] = lit; S[2] = dist; S[3] = ml; }
}

// one wave per block: exclusive sums over the segments, the decisions of :559-561 and :586-588, the header
__global__ __launch_bounds__(64) void knz_lzs_emit_offsets_kernel(LzSegArgs g, uint32_t* segsum, uint32_t* blkout) {
    const LzArgs& a = g.pa.a;
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    if (g.blk_state[b] != 1) return;
    const int count = (int)a.in_len[b];
    int srcEnd, maxDist, minMatch; uint32_t hf; bool decline;
    knz_lzs_geom(a, b, count, srcEnd, maxDist, minMatch, hf, decline);
    const uint32_t ns = srcEnd > 0 ? ((uint32_t)srcEnd + g.seg_size - 1) / g.seg_size : 0;
    uint32_t c[4] = {0, 0, 0, 0};
    for (uint32_t s0 = 0; s0 < ns; s0 += 64) {
        const uint32_t s = s0 + lane;
        uint32_t* S = segsum + 4 * ((size_t)b * g.segs + s);
        uint32_t v[4] = {0, 0, 0, 0};
        if (s < ns) { v[0] = S[0]; v[1] = S[1]; v[2] = S[2]; v[3] = S[3]; }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t incl = wave_scan_incl(v[q]);
            if (s < ns) S[q] = c[q] + incl - v[q];
            c[q] += wave_shfl(incl, 63);
        }
    }
    if (lane != 0) return;
    const uint32_t anchor = g.blk_flags[4 * b];
    const uint32_t litLen = (uint32_t)count - anchor;
    uint32_t dstIdx = 13u + c[1];
    const uint32_t tkIdx = c[0], mIdx = c[2], mLenIdx = c[3];
    const uint32_t tkCap = (uint32_t)(count / 5 > 256 ? count / 5 : 256);
    uint32_t* O = blkout + 8 * (size_t)b;
    int status = 1;
    if (tkIdx >= tkCap) { g.blk_state[b] = 2; return; }                       // the reference's token buffer would overflow: the one-wave kernel reports it
    if ((uint64_t)dstIdx + litLen + tkIdx + mIdx >= (uint64_t)count) status = 0;   // "no compression" (:559-561)
    else {
        const uint32_t litStart = dstIdx;                                     // the last literals (:563-573)
        dstIdx += (litLen >= 7 ? knz_lzs_len_size(litLen - 7) : 0u) + litLen;
        const uint32_t total = dstIdx + (tkIdx + 1) + mIdx + mLenIdx;
        O[0] = litStart; O[1] = dstIdx; O[2] = dstIdx + tkIdx + 1; O[3] = dstIdx + tkIdx + 1 + mIdx; O[4] = anchor; O[5] = litLen; O[6] = tkIdx;
        uint8_t* dst = (uint8_t*)a.out_ptr[b];
        const uint32_t v0 = dstIdx, v1 = tkIdx + 1, v2 = mIdx;
        for (int k = 0; k < 4; k++) { dst[k] = (uint8_t)(v0 >> (8 * k)); dst[4 + k] = (uint8_t)(v1 >> (8 * k)); dst[8 + k] = (uint8_t)(v2 >> (8 * k)); }
        dst[12] = (uint8_t)hf;
        if (total > (uint32_t)(count - count / 100)) status = 0;             // :586-588
        a.out_len[b] = status == 1 ? total : 0;
    }
    O[7] = (uint32_t)status;
    a.ok[b] = status;
    if (status != 1) a.out_len[b] = 0;
}

// per segment: the bytes of its tokens at their places; grid (segs, nblocks). Tiles of 256 tokens: sizes, prefix sums, then the token /
// distance / length bytes by one thread per token and the literals by all threads over the tile's byte range.
__global__ __launch_bounds__(256) void knz_lzs_emit_write_kernel(LzSegArgs g, const uint32_t* segsum, const uint32_t* blkout) {
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_off[257], s_src[256], s_pre[257];
    const LzArgs& a = g.pa.a;
    const uint32_t b = blockIdx.y, s = blockIdx.x, tid = threadIdx.x;
    if (g.blk_state[b] != 1) return;
    const uint32_t* O = blkout + 8 * (size_t)b;
    if (O[7] != 1) return;
    const int count = (int)a.in_len[b];
    int srcEnd, maxDist, minMatch; uint32_t hf; bool decline;
    knz_lzs_geom(a, b, count, srcEnd, maxDist, minMatch, hf, decline);
    const size_t si = (size_t)b * g.segs + s;
    const uint32_t n = knz_lzs_live(g, b, s, srcEnd) ? g.ntok[si] : 0u;
    if (n == 0) return;
    const uint4* tk = g.tok + si * g.tok_cap;
    const uint32_t* S = segsum + 4 * si;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    uint32_t tkPos = O[1] + S[0], litPos = 13u + S[1], mPos = O[2] + S[2], mlPos = O[3] + S[3];
    for (uint32_t i0 = 0; i0 < n; i0 += 256) {
        const uint32_t i = i0 + tid;
        uint4 t; t.x = t.y = t.z = t.w = 0;
        LzsSizes z; z.lit = z.dist = z.ml = 0; z.special = false;
        if (i < n) { t = tk[i]; z = knz_lzs_sizes(t, (uint32_t)minMatch); }
        const uint32_t eLit = knz_lzi_wg_scan_excl(z.lit, s_w), eDist = knz_lzi_wg_scan_excl(z.dist, s_w), eMl = knz_lzi_wg_scan_excl(z.ml, s_w);
        const uint32_t litLen = t.y, bestLen = t.z & 0xFFFFFFu, fl = t.z >> 24, dist = t.w;
        const uint32_t extSz = litLen >= 7 ? knz_lzs_len_size(litLen - 7) : 0u;
        s_off[tid] = litPos + eLit + extSz; s_src[tid] = t.x;
        if (i < n) {
            const uint32_t mLen = bestLen - (uint32_t)minMatch, th = fl < 8 ? 3u : 7u;
            const uint32_t token = fl + (mLen >= th ? th : mLen);
            dst[tkPos + i] = (uint8_t)((litLen >= 7 ? (7u << 5) : (litLen << 5)) | token);
            if (litLen >= 7) (void)knz_lz_emit_length(dst + litPos + eLit, (int)(litLen - 7), true);
            uint8_t* m = dst + mPos + eDist;
            if (fl == 0x18) { m[0] = (uint8_t)(dist >> 16); m[1] = (uint8_t)(dist >> 8); m[2] = (uint8_t)dist; }
            else if (fl == 0x10) { m[0] = (uint8_t)(dist >> 8); m[1] = (uint8_t)dist; }
            else if (fl == 0x08) m[0] = (uint8_t)dist;
            if (mLen >= th) (void)knz_lz_emit_length(dst + mlPos + eMl, (int)(mLen - th), true);
        }
        const uint32_t nt = min(256u, n - i0);
        // totals of the tile (all threads took part in the scans above)
        __shared__ uint32_t s_tot[3];
        if (tid == 255) { s_tot[0] = eLit + z.lit; s_tot[1] = eDist + z.dist; s_tot[2] = eMl + z.ml; }
        __syncthreads();
        // literals of the tile, one thread per BYTE: the literal runs of a segment are short (2-3 bytes on average), so a wave that walks its 64 tokens one
        // after the other spends its time on loop overhead; a thread finds the token its byte belongs to by a search over the tile's run starts
        {
            const uint32_t ePlain = knz_lzi_wg_scan_excl(i < n ? litLen : 0u, s_w);       // starts of the runs, counted in literal bytes alone
            s_pre[tid] = ePlain;
            if (tid == 255) s_pre[256] = ePlain + (i < n ? litLen : 0u);
            __syncthreads();
            const uint32_t total = s_pre[256];
            for (uint32_t o = tid; o < total; o += 256) {
                uint32_t lo = 0, hi = nt;                                                    // largest t with s_pre[t] <= o (runs of length 0 share a start: the last one wins)
                while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (s_pre[mid] <= o) lo = mid; else hi = mid; }
                const uint32_t k = o - s_pre[lo];
                dst[s_off[lo] + k] = src[s_src[lo] + k];
            }
        }
        litPos += s_tot[0]; mPos += s_tot[1]; mlPos += s_tot[2];
        __syncthreads();
    }
}

// the last literals of a block (:563-573): one workgroup per block
__global__ __launch_bounds__(256) void knz_lzs_emit_tail_kernel(LzSegArgs g, const uint32_t* blkout) {
    const LzArgs& a = g.pa.a;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    if (g.blk_state[b] != 1) return;
    const uint32_t* O = blkout + 8 * (size_t)b;
    if (O[7] != 1) return;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const uint32_t anchor = O[4], litLen = O[5];
    uint32_t p = O[0];
    const uint32_t ext = litLen >= 7 ? knz_lzs_len_size(litLen - 7) : 0u;
    if (tid == 0) {
        dst[O[1] + O[6]] = (uint8_t)(litLen >= 7 ? (7u << 5) : (litLen << 5));
        if (litLen >= 7) (void)knz_lz_emit_length(dst + p, (int)(litLen - 7), true);
    }
    p += ext;
    for (uint32_t o = tid; o < litLen; o += 256) dst[p + o] = src[anchor + o];
}

// blocks the one-wave kernel has to take (mask for its `active` argument)
__global__ __launch_bounds__(64) void knz_lzs_serial_mask_kernel(LzSegArgs g, uint8_t* mask) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= g.pa.a.nblocks) return;
    mask[b] = (g.blk_state[b] == 2 || g.blk_state[b] == 4) ? 1 : 0;
}
