// ZRLT^-1 -> RANK^-1 of a block as ONE chain that starts under the order-1 rANS decoder (decode of `-t ...RANK+ZRLT -e ANS1`).
//
// Why: both the order-1 rANS decoder (ans1.hip: one wave per 4 MiB chunk, ~150 ms) and the inverse RANK (rank_inv.hip: one wave per block,
// ~0.2-0.4 s) are chains by format, and until round 4 they ran one after the other. The rANS decoder advances the four quarters of a chunk in
// lock-step, so the FIRST quarter of the ZRLT stream is complete exactly when the chunk is. This kernel (one wave per block, launched on a
// second stream while the decoder runs) polls the decoder's progress word, expands what is there (ZRLT.go:142-225, wave-parallel: 32 bytes per
// lane) into ranks and runs the inverse RANK chain (SBRT.go:180-226) over them at the producer's pace; once the chunk is done it runs free over
// the other three quarters. A quarter of the chain's time leaves the critical path of the step.
//
// Hand-over between the two kernels (different workgroups, possibly different XCDs): the decoder stores its bytes, fences at agent scope and
// publishes its step count (knz_publish64); this wave reads the count (knz_poll64), fences (acquire, agent scope) and reads the bytes.
// Memory: ZRLT stream = the decoder's output in region 1; ranks go to region 3 (zero-filled before: only literals are scattered); the decoded
// symbols go to region 2, where the regular inverse RANK of a block that took two stages would have left them two stages later.
// Anything this wave does not like (a stream no encoder writes: 0xFF 0xFF, a run of more than 31 digits, output beyond the block's region, a
// producer that does not move for 250 ms) leaves piped[b] = 0: the regular stage kernels then take the block from the decoder's untouched output.
#pragma once

struct RankPipeArgs {
    uint32_t nblocks;
    uint32_t chunks_per_block;
    const uint32_t* info;             // [nslots * 8] the rANS chunk headers ({mode, st0..st3, lr}): mode 3 = coded chunk
    const uint64_t* progress;         // [nslots] steps the decoder of the chunk has stored (its first-quarter bytes), KNZ_PIPE_DONE when the chunk is complete
    uint64_t* cur_ptr;                // [nblocks] in: the ZRLT stream (decoder output); out: the decoded symbols
    uint32_t* cur_len;                // [nblocks] in: its length; out: the block's length behind the two stages
    const uint8_t* skip;              // [nblocks] skip flags of the block header
    uint8_t* side;                    // [nblocks] which region the block is in (1 / 2)
    const int32_t* blk_status;
    uint8_t* piped;                   // [nblocks] out: 1 = both stages are done for this block
    uint64_t ranks_base, out_base, stride;   // region 3 (ranks), region 2 (symbols), bytes per block
    uint32_t out_cap;
    uint32_t zrlt_stage, rank_stage;  // positions of the two transforms in the sequence (their skip bits)
    uint32_t mode;                    // bit 8: force the three-register chain; bits 12..: packed/unpacked cut in rows (tests)
    const uint8_t* group;             // [nblocks] or null: which of the two launches takes the block (the long chains go in a launch of their own, see decode_batch)
    uint32_t group_sel;
};
#define KNZ_PIPE_DONE 0xFFFFFFFFFFFFFFFFull
#define KNZ_PIPE_GIVE_UP_TICKS 25000000ull   // 250 ms of the 100 MHz counter
#define KNZ_PIPE_PIECE 2048u          // input bytes per expansion step: 32 per lane

// one piece [lo, hi) of the ZRLT stream (hi - lo <= 2048, lo a multiple of 2048): returns the ranks it produced, or 0xFFFFFFFF to decline.
// carry: (cv, cL) = the digits of a run that the piece before left open (cv = 1, cL = 0: none), cprev = the byte in front of lo is an escape.
// WRITE = false: count only.
struct ZiLane {
    uint32_t w[9];                    // the lane's 32 bytes + the 4 behind them
    uint32_t prevByte;
};

template <bool WRITE>
__device__ __forceinline__ uint32_t knz_zi_lane_pass(const ZiLane& z, uint32_t p0, uint32_t hi, uint32_t m, uint32_t cap, uint32_t cv, uint32_t cL, uint8_t* dst,
                                                    uint32_t off, uint32_t& tv, uint32_t& tL, uint32_t& lastc, bool& bad) {
    uint32_t v = cv, L = cL, sz = 0;
    bool payload = z.prevByte == 0xFFu;                      // the byte in front of the lane's first is an escape => the first byte is its payload
#pragma unroll
    for (uint32_t j = 0; j < 32; j++) {                      // (fully unrolled: the bytes are picked out of registers with constant shifts)
        const uint32_t i = p0 + j;
        const uint32_t c = (z.w[j >> 2] >> (8 * (j & 3))) & 0xFFu;
        const uint32_t nx = (z.w[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 0xFFu;
        if (i < hi) {
            lastc = c;
            if (payload) {                                   // 0xFF x => 0xFE + x (ZRLT.go:196-204)
                if (c == 0xFFu) bad = true;                  // (no encoder writes 0xFF 0xFF)
                if (WRITE) dst[off + sz] = (uint8_t)(0xFEu + c);
                sz += 1;
                payload = false;
            } else if (c <= 1u) {                            // a digit of a zero run (:165-180)
                v = (v << 1) | c;
                L++;
                const bool more = (i + 1 < m) && nx <= 1u;
                if (!more) {                                 // the run ends here: v - 1 zeros (already there: the region is zero-filled)
                    if (L > 31u || v - 1u > cap) bad = true;
                    sz += v - 1u;
                    v = 1u; L = 0;
                }
            } else if (c == 0xFFu) {
                payload = true;
            } else {
                if (WRITE) dst[off + sz] = (uint8_t)(c - 1u);
                sz += 1;
            }
            if (sz > cap) { bad = true; sz = 0; }
        }
    }
    tv = v; tL = L;
    return sz;
}

// expands [lo, hi) and returns the number of ranks (0xFFFFFFFF: decline). State in / out: cv, cL (open run), lastByte (the byte at hi - 1).
__device__ __forceinline__ uint32_t knz_zi_piece(const uint8_t* src, uint32_t lo, uint32_t hi, uint32_t m, uint8_t* ranks, uint32_t outPos, uint32_t cap,
                                                 uint32_t& cv, uint32_t& cL, uint32_t& lastByte, int lane) {
    const uint32_t p0 = lo + 32u * (uint32_t)lane;
    ZiLane z;
    const uint4* sp = (const uint4*)(src + p0);
    const bool mine = p0 < hi;
    uint4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    uint32_t nxt = 0x02020202u;
    if (mine) {                                              // (the block regions are 16-byte aligned and padded by 64 bytes: whole vectors are always inside)
        a = sp[0]; b = sp[1];
        nxt = *(const uint32_t*)(src + p0 + 32);
    }
    z.w[0] = a.x; z.w[1] = a.y; z.w[2] = a.z; z.w[3] = a.w; z.w[4] = b.x; z.w[5] = b.y; z.w[6] = b.z; z.w[7] = b.w; z.w[8] = nxt;
    const uint32_t myLast = z.w[7] >> 24;
    const uint32_t upLast = wave_shfl(myLast, (lane + 63) & 63);
    z.prevByte = lane == 0 ? lastByte : upLast;
    // pass 1: what each lane leaves open (no carry needed: a lane of 32 digits is not a stream an encoder writes)
    uint32_t tv = 1, tL = 0, lc = 0;
    bool bad = false;
    if (mine) (void)knz_zi_lane_pass<false>(z, p0, hi, m, cap, 1u, 0u, nullptr, 0u, tv, tL, lc, bad);
    if (mine && tL >= 32u) bad = true;
    uint32_t inV = wave_shfl(tv, (lane + 63) & 63), inL = wave_shfl(tL, (lane + 63) & 63);
    if (lane == 0) { inV = cv; inL = cL; }
    // pass 2: sizes with the carried digits, pass 3: the literals to their places
    uint32_t t2v = 1, t2L = 0;
    const uint32_t sz = mine ? knz_zi_lane_pass<false>(z, p0, hi, m, cap, inV, inL, nullptr, 0u, t2v, t2L, lc, bad) : 0u;
    if (wave_ballot(bad) != 0) return 0xFFFFFFFFu;
    const uint32_t incl = wave_scan_incl(sz);
    const uint32_t total = wave_bcast(incl, 63);
    if ((uint64_t)outPos + total > cap) return 0xFFFFFFFFu;                    // (ZRLT.go:178-180, 211-216: the regular stage reports it)
    if (mine) (void)knz_zi_lane_pass<true>(z, p0, hi, m, cap, inV, inL, ranks, outPos + incl - sz, t2v, t2L, lc, bad);
    // what the piece leaves open = the last lane that has bytes
    const int lastLane = (int)((hi - lo - 1u) >> 5);
    cv = wave_readlane(t2v, (uint32_t)lastLane); cL = wave_readlane(t2L, (uint32_t)lastLane);
    lastByte = wave_readlane(lc, (uint32_t)lastLane);
    return total;
}

#if defined(KNZ_MEASURE) && !defined(KNZ_HIP_EMU)
__device__ unsigned long long g_knz_pipe_ticks[1024][8];      // diagnostics (KNZ_RANK_PROF): per block, 100 MHz ticks {chain waits, expander total, chain, total, first data, producer done seen}, m, ranks
#define KNZ_PIPE_T(...) __VA_ARGS__
#else
#define KNZ_PIPE_T(...)
#endif
// Two waves per block: wave 1 expands the ZRLT stream as the decoder hands it over and tells wave 0 (LDS) how many ranks are there; wave 0 is the
// inverse RANK chain. The expansion (~10 ns per input byte) is off the chain's path: the waves sit on different SIMDs of the CU.
#define KNZ_PIPE_RUN 0u
#define KNZ_PIPE_END 1u
#define KNZ_PIPE_DECLINED 2u
template <int MODE, int FLAGS>
__global__ __launch_bounds__(128) void knz_zrlti_rank_pipe_kernel(RankPipeArgs a) {
    __shared__ uint32_t s_ready;                                // ranks expanded so far (written by wave 1, read by wave 0)
    __shared__ uint32_t s_state;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x;
    if (a.group != nullptr && a.group[b] != a.group_sel) return;
    // --- does this block take the fused path at all?
    const uint32_t m = a.cur_len[b];
    const uint32_t K = a.chunks_per_block;
    bool take = a.blk_status[b] == 0 && m != 0 && a.side[b] == 1 &&
                !(a.skip[b] & (1u << (7 - a.zrlt_stage))) && !(a.skip[b] & (1u << (7 - a.rank_stage)));
    const uint32_t nch = take ? (m + KNZ_ANS1_CHUNK - 1) / KNZ_ANS1_CHUNK : 0;
    for (uint32_t k = 0; k < nch && take; k++) if (a.info[((size_t)b * K + k) * 8] != 3u) take = false;     // raw chunks: the regular path
    if (!take) return;
    const uint8_t* src = (const uint8_t*)a.cur_ptr[b];
    uint8_t* ranks = (uint8_t*)(a.ranks_base + (uint64_t)b * a.stride);
    uint8_t* dst = (uint8_t*)(a.out_base + (uint64_t)b * a.stride);
    const uint32_t cap = a.out_cap;
    if (threadIdx.x == 0) { s_ready = 0; s_state = KNZ_PIPE_RUN; }
    __syncthreads();
    KNZ_PIPE_T(const unsigned long long tStart = KNZ_RANK_NOW();)
    if (wave == 1) {
        // ---------------------------------------------------------------- the expander
        uint32_t inPos = 0, outPos = 0, cv = 1, cL = 0, lastByte = 0, fenced = 0, idle = 0;
        uint64_t idleSince = 0;
        KNZ_PIPE_T(unsigned long long tFirst = 0, tDone = 0;)
        while (inPos < m) {
            // contiguous prefix of the stream that is there: whole chunks that are done + the first quarter of the one in flight
            uint32_t avail = 0;
            for (uint32_t k = 0; k < nch; k++) {
                const uint32_t nk = min((uint32_t)KNZ_ANS1_CHUNK, m - k * (uint32_t)KNZ_ANS1_CHUNK);
                const uint64_t p = knz_poll64(a.progress + (size_t)b * K + k);
                if (p == KNZ_PIPE_DONE) { avail += nk; continue; }
                avail += (uint32_t)min(p, (uint64_t)((nk & ~3u) >> 2));
                break;
            }
            if (avail > fenced) { agent_fence_acquire(); fenced = avail; }
            KNZ_PIPE_T(if (tFirst == 0 && fenced != 0) tFirst = KNZ_RANK_NOW() - tStart; if (tDone == 0 && fenced == m) tDone = KNZ_RANK_NOW() - tStart;)
            const uint32_t limit = fenced == m ? m : (fenced ? fenced - 1u : 0u);          // (a digit's run ends where the NEXT byte says so)
            bool moved = false;
            uint32_t budget = 16;                                                          // pieces per hand-over to the chain
            while (inPos < limit && budget-- != 0) {
                const uint32_t hi = min(limit, (inPos & ~(KNZ_PIPE_PIECE - 1u)) + KNZ_PIPE_PIECE);
                if (hi - inPos < KNZ_PIPE_PIECE && hi != m) break;                          // wait for the whole piece unless the stream ends in it
                const uint32_t got = knz_zi_piece(src, inPos, hi, m, ranks, outPos, cap, cv, cL, lastByte, lane);
                if (got == 0xFFFFFFFFu) { if (lane == 0) s_state = KNZ_PIPE_DECLINED; return; }   // piped[b] stays 0: the regular stage kernels take the block
                outPos += got;
                inPos = hi;
                moved = true;
            }
            if (moved) {
                agent_fence_release();                                                     // the literals have landed in L2 before the chain (which reads them back through the scalar cache) is told
                if (lane == 0) s_ready = outPos;
                idle = 0;
            } else {
                if (idle++ == 0) idleSince = knz_realtime();
                wg_spin_pause();
                // a producer that has not moved for a quarter of a second of wall clock (its workgroup is queued behind a full device): hand the block back
                if ((idle & 63u) == 0 && knz_realtime() - idleSince > KNZ_PIPE_GIVE_UP_TICKS) { if (lane == 0) s_state = KNZ_PIPE_DECLINED; return; }
            }
        }
        agent_fence_release();
        if (lane == 0) { s_ready = outPos; s_state = KNZ_PIPE_END; }
        KNZ_PIPE_T(if (lane == 0 && b < 1024u) { unsigned long long* t = g_knz_pipe_ticks[b]; t[1] = KNZ_RANK_NOW() - tStart; t[4] = tFirst; t[5] = tDone; t[6] = m; t[7] = outPos; })
        return;
    }
    // -------------------------------------------------------------------- the chain
    constexpr bool WIDE = (FLAGS & 8) != 0;
    constexpr int XP = (FLAGS >> 4) & 7;
    const bool allowPacked = !(a.mode & 0x100);
    const uint32_t cutRows = a.mode >> 12;
    const uint32_t cut = allowPacked ? (cutRows ? min(64u * cutRows, KNZ_RANK_PACKED_TIMES) : KNZ_RANK_PACKED_TIMES) : 0u;
    RankChainV<MODE, true, WIDE, XP> c;
    RankChainV<MODE, false, WIDE, XP> u;
    c.init_identity(lane);
    u.init_identity(lane);
    bool unpacked = !allowPacked;
    uint32_t chainPos = 0, outPos = 0;
    KNZ_PIPE_T(unsigned long long tW = 0, tC = 0; unsigned long long tMark = tStart;)
    for (;;) {
        const uint32_t state = wave_uniform(*(volatile uint32_t*)&s_state);                // (read before the count: END is published behind the last count)
        const uint32_t total = wave_uniform(*(volatile uint32_t*)&s_ready);
        if (state == KNZ_PIPE_DECLINED) return;
        const uint32_t ready = state == KNZ_PIPE_END ? total : (total & ~63u);
        if (ready > chainPos) {
            KNZ_PIPE_T({ const unsigned long long now = KNZ_RANK_NOW(); tW += now - tMark; tMark = now; })
            agent_fence_acquire();
            knz_scalar_cache_inv();                                                        // the rows come back through the scalar cache
            uint32_t end = min(ready, chainPos + (1u << 17));
            if (!unpacked) {
                if (chainPos < cut) {
                    end = min(end, cut);
                    knz_rank_chain_range_v<MODE, true, WIDE, XP>(c, ranks, dst, chainPos, end, lane);
                    chainPos = end;
                }
                if (chainPos >= cut) {                          // the packed form holds up to time 2^23 (rank_inv.hip): from there the three-register form
                    u.lane = lane;
#pragma unroll
                    for (int k = 0; k < 4; k++) { u.e[k] = c.e[k] & 0xFFu; u.p[k] = c.e[k] >> 8; u.q[k] = c.q[k]; }
                    u.qp = 0x7FFFFFFF;
                    u.vff = c.vff;
                    u.refresh();
                    unpacked = true;
                }
            } else {
                knz_rank_chain_range_v<MODE, false, WIDE, XP>(u, ranks, dst, chainPos, end, lane);
                chainPos = end;
            }
            KNZ_PIPE_T({ const unsigned long long now = KNZ_RANK_NOW(); tC += now - tMark; tMark = now; })
        } else if (state == KNZ_PIPE_END) { outPos = total; break; }
        else wave_spin_pause();
    }
    if ((uint64_t)outPos > cap) return;
    KNZ_PIPE_T(if (lane == 0 && b < 1024u) { unsigned long long* t = g_knz_pipe_ticks[b]; t[0] = tW; t[2] = tC; t[3] = KNZ_RANK_NOW() - tStart; })
    if (lane == 0) {
        a.cur_ptr[b] = (uint64_t)dst;
        a.cur_len[b] = outPos;
        a.side[b] = 2;
        a.piped[b] = 1;
    }
}
