// (Round 3: the default device build runs the groups of sixteen accesses of the packed RANK chain through the hand-written blocks of
// rank_inv_asm.h; what follows is the C++ form of the same steps: the emulator's, the measurement variants', MTFT's and the
// three-register form's.)
// Inverse RANK / MTFT as ONE chain per block (SBRT.Inverse, v2/transform/SBRT.go:180-226), rebuilt around the two things that
// bound a lone wave on gfx950: the number of instructions per symbol and the number of VALU <-> SALU hand-overs on the
// loop-carried path (each one costs a pipeline drain: ~15 cycles per instruction were measured on the round-1 form, whose
// step went v_readlane -> SALU -> v_cmp -> s_bcnt -> VALU masks -> 6 v_cndmask).
//
// The list stays in registers (rank l of the first 64 in lane l) and stays sorted by q. For an access of rank r at time i
//     qc = (i + p[r]) >> 1                                  (SBRT.go:211, mode RANK; MTFT: qc = i)
// every lane can decide on its own what it holds afterwards, WITHOUT the landing rank j ever being computed:
//     q[l]   >  qc           -> lane keeps its entry                      (l <  j)
//     q[l-1] >  qc >= q[l]   -> lane takes the accessed entry             (l == j)
//     otherwise, l <= r      -> lane takes the entry of lane l-1          (j <  l <= r)
// i.e. two v_cmp against the uniform qc (one on q, one on the lane-1 copy of q that a DPP move keeps up to date) and two
// v_cndmask per register. One hand-over to the scalar unit (v_readlane of the accessed entry -> qc) and one back per step, no
// popcount, no lane masks built from j. Symbol and last access time share a register (p << 8 | sym) while times fit
// (blocks <= 8 MiB: the BASELINE block size of this pipeline); larger blocks use a three-register form of the same step.
//
// Input ranks are fetched with scalar loads (s_load_dwordx4: the address is wave-uniform and the bytes are read-only for this
// kernel), 16 ranks per instruction and one group ahead of the chain; words of four zero ranks (the bulk behind a BWT) cost
// one scalar compare. Decoded symbols are packed four to a word in SGPRs, parked in lane (word & 63) of one VGPR and stored
// 256 bytes at a time. Ranks >= 64 (rare behind a BWT) take the four-register path of the round-1 kernel.
#pragma once

// FLAGS (variants kept for measurement, see docs/HISTORY.md section 4): bit 0: an isolated rank 0 takes a branch to a shorter step;
// bit 1: lanes above r are kept by a per-lane threshold (v_cmp + v_cndmask) instead of an EXEC mask around the selects

template <int MODE, bool PACKED, int FLAGS>
struct RankChain {
    // registers k = 0..3 hold ranks 64k .. 64k+63. PACKED: e = p << 8 | sym ; else e = sym and p separately
    uint32_t e[4], p[4];
    int q[4];
    uint32_t ep, pp;      // lane-1 copies of register 0 (lane 0: don't care)
    int qp;               // lane-1 copy of q[0], lane 0: INT_MAX ("there is always something above rank 0")
    int lane;

    __device__ __forceinline__ void refresh() {
        ep = wave_shr1(e[0]);
        if (!PACKED) pp = wave_shr1(p[0]);
        qp = (int)wave_shr1_old((uint32_t)q[0], 0x7FFFFFFFu);
    }
    __device__ __forceinline__ void init_identity(int l) {
        lane = l;
#pragma unroll
        for (int k = 0; k < 4; k++) { e[k] = 64u * (uint32_t)k + (uint32_t)l; p[k] = 0; q[k] = 0; }   // p = 0: packed e is the symbol itself
        refresh();
    }
    static __device__ __forceinline__ int qf(uint32_t i, uint32_t pc) { return MODE == 1 ? (int)i : (int)((i + pc) >> 1); }

    // rank r <= 63 at time i; returns the symbol
    __device__ __forceinline__ uint32_t step_low(uint32_t r, uint32_t i) {
        const uint32_t se = wave_readlane(e[0], r);
        uint32_t sym, snew, spn = i;
        int qc;
        if (PACKED) { sym = se & 0xFFu; qc = MODE == 1 ? (int)i : (int)(((i << 8) + se) >> 9); snew = (i << 8) | sym; }   // (i<<8) + (p<<8|sym) < 2^32 for i, p < 2^23
        else { sym = se; qc = qf(i, wave_readlane(p[0], r)); snew = sym; }
        if (FLAGS & 2) {
            const int qcl = (uint32_t)lane > r ? -1 : qc;                          // q >= 0: lanes above r always keep
            const bool keep = q[0] > qcl, ins = qp > qc;
            e[0] = keep ? e[0] : (ins ? snew : ep);
            if (!PACKED) p[0] = keep ? p[0] : (ins ? spn : pp);
            q[0] = keep ? q[0] : (ins ? qc : qp);
        } else if ((uint32_t)lane <= r) {
            const bool keep = q[0] > qc, ins = qp > qc;
            e[0] = keep ? e[0] : (ins ? snew : ep);
            if (!PACKED) p[0] = keep ? p[0] : (ins ? spn : pp);
            q[0] = keep ? q[0] : (ins ? qc : qp);
        }
        refresh();
        return sym;
    }
    // rank 0 (the symbol stays on top: nothing above it)
    __device__ __forceinline__ uint32_t step_top(uint32_t i) {
        const uint32_t se = wave_readlane(e[0], 0);
        uint32_t sym;
        int qc;
        if (PACKED) { sym = se & 0xFFu; qc = MODE == 1 ? (int)i : (int)(((i << 8) + se) >> 9); e[0] = wave_writelane0(e[0], (i << 8) | sym); }
        else { sym = se; qc = qf(i, wave_readlane(p[0], 0)); p[0] = wave_writelane0(p[0], i); }
        q[0] = (int)wave_writelane0((uint32_t)q[0], (uint32_t)qc);
        refresh();
        return sym;
    }
    // `count` >= 2 consecutive ranks 0, the last one at time i
    __device__ __forceinline__ uint32_t run_top(uint32_t i, uint32_t count) {
        const uint32_t se = wave_readlane(e[0], 0);
        const uint32_t sym = PACKED ? (se & 0xFFu) : se;
        if (PACKED) e[0] = wave_writelane0(e[0], (i << 8) | sym); else p[0] = wave_writelane0(p[0], i);
        q[0] = (int)wave_writelane0((uint32_t)q[0], (uint32_t)qf(i, i - 1));
        refresh();
        return sym;
    }
    // rare: rank 64..255. Landing rank by popcount over the four registers, one-lane shifts with the carry between registers.
    __device__ __forceinline__ uint32_t step_high(uint32_t r, uint32_t i) {
        const uint32_t kr = r >> 6, l = r & 63;
        const uint32_t se = kr == 1 ? wave_readlane(e[1], l) : (kr == 2 ? wave_readlane(e[2], l) : wave_readlane(e[3], l));
        uint32_t sym, snew, pc;
        if (PACKED) { sym = se & 0xFFu; pc = se >> 8; snew = (i << 8) | sym; }
        else { sym = se; snew = se; pc = kr == 1 ? wave_readlane(p[1], l) : (kr == 2 ? wave_readlane(p[2], l) : wave_readlane(p[3], l)); }
        const int qc = qf(i, pc);
        uint32_t j = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) j += (uint32_t)__popcll(wave_ballot(q[k] > qc));
#pragma unroll
        for (int k = 3; k >= 0; k--) {                                            // high registers first: register k-1 is still old
            const uint32_t x = 64u * (uint32_t)k + (uint32_t)lane;
            const bool moved = x - j - 1u < r - j;
            const bool ins = x == j;
            uint32_t es = wave_shr1(e[k]), ps = wave_shr1(p[k]), qs = wave_shr1((uint32_t)q[k]);
            if (k > 0) {
                const uint32_t e63 = wave_bcast(e[k > 0 ? k - 1 : 0], 63), p63 = wave_bcast(p[k > 0 ? k - 1 : 0], 63), q63 = wave_bcast((uint32_t)q[k > 0 ? k - 1 : 0], 63);
                if (lane == 0) { es = e63; ps = p63; qs = q63; }
            }
            e[k] = ins ? snew : (moved ? es : e[k]);
            if (!PACKED) p[k] = ins ? i : (moved ? ps : p[k]);
            q[k] = ins ? qc : (moved ? (int)qs : q[k]);
        }
        refresh();
        return sym;
    }
    __device__ __forceinline__ uint32_t step_any(uint32_t r, uint32_t i) {
        if (r == 0) return step_top(i);
        if (r < 64) return step_low(r, i);
        return step_high(r, i);
    }
    // four ranks of one input word, first at time i; returns the four symbols. A word with a rank >= 64 in it (rare behind a
    // BWT) goes through the general step, so that the unrolled part carries no trace of the four-register path.
    __device__ __forceinline__ uint32_t word(uint32_t w, uint32_t i) {
        if (w == 0) return run_top(i + 3, 4) * 0x01010101u;
        uint32_t out = 0;
        if (w & 0xC0C0C0C0u) {
#pragma nounroll
            for (uint32_t u = 0; u < 4; u++) out |= step_any((w >> (8 * u)) & 0xFFu, i + u) << (8 * u);
            return out;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t r = (w >> (8 * u)) & 0xFFu;
            out |= ((FLAGS & 1) && r == 0 ? step_top(i + (uint32_t)u) : step_low(r, i + (uint32_t)u)) << (8 * u);
        }
        return out;
    }
};

template <int MODE, bool PACKED, int FLAGS>
__device__ __forceinline__ void knz_rank_chain_block(const uint8_t* src, uint8_t* dst, uint32_t n, int lane) {
    RankChain<MODE, PACKED, FLAGS> c;
    c.init_identity(lane);
    uint32_t i0 = 0;
    if (((((uintptr_t)src) | ((uintptr_t)dst)) & 3) == 0) {                     // (the pipeline's own regions are 16-byte aligned)
        const uint32_t nw = n >> 2;                                             // whole words
        uint32_t outv = 0, k = 0;
        knz_u32x4 nxt = {0, 0, 0, 0};
        if (nw >= 4) nxt = wave_sload_u32x4(src);
        while (k + 4 <= nw) {
            knz_u32x4 cur = nxt;
            if (k + 8 <= nw) nxt = wave_sload_u32x4(src + 4 * (size_t)(k + 4)); // one group ahead of the chain
            const uint32_t sel = (uint32_t)lane - (k & 63);                     // k is a multiple of 4: the group stays inside one 64-word row
            if ((cur.x | cur.y | cur.z | cur.w) == 0) {
                const uint32_t o = c.run_top(4 * k + 15, 16) * 0x01010101u;
                outv = sel < 4 ? o : outv;
            } else {
#pragma nounroll
                for (uint32_t wi = 0; wi < 4; wi++) {
                    const uint32_t o = c.word(cur.x, 4 * (k + wi));
                    outv = sel == wi ? o : outv;
                    cur.x = cur.y; cur.y = cur.z; cur.z = cur.w;
                }
            }
            k += 4;
            if ((k & 63) == 0) ((uint32_t*)dst)[k - 64 + lane] = outv;          // 256 decoded bytes, coalesced
        }
        if ((k & 63) != 0 && (uint32_t)lane < (k & 63)) ((uint32_t*)dst)[(k & ~63u) + lane] = outv;
        i0 = 4 * k;
    }
    for (uint32_t i = i0; i < n; i++) {                                         // the last < 16 ranks (or unaligned buffers): byte by byte
        const uint32_t s = c.step_any(wave_uniform(src[i]), i);
        if (lane == 0) dst[i] = (uint8_t)s;
    }
}

// ---- the same chain with the uniform arithmetic on the vector ALU -----------------------------------------------------
// Measured on one wave of an MI355X (tools/gpu/lat_bench.hip, profiles/r02_lone_wave_latencies.md): every instruction of a
// lone wave costs ~2.5 ns whatever unit it runs on (5.75 clocks at 2.39 GHz), an SGPR written by the VALU and read by the
// SALU adds ~6 ns, a taken branch ~10 ns. So the step is built to be SHORT and to stay on the vector ALU: the entry comes
// back from v_readlane in an SGPR and is consumed by VALU instructions only (qc = (i8 + e) >> 9 and the new entry
// (e & 0xFF) | i8 as wave-uniform VGPRs, v_and_or_b32), lanes above r are protected by replacing their q with INT_MAX
// before the compare (off the loop-carried path: r is known early), the lane-1 copy of q keeps its +inf in lane 0 by being
// shifted in place (DPP without bound_ctrl leaves lane 0 alone), and the decoded byte leaves through one v_writelane with a
// compile-time lane (16 symbols per VGPR, stored by 16 lanes). ~17 instructions per symbol, no VALU -> SALU hand-over.
// Round 3 (XP & 4, the default): the lane-1 copies of (q, e) are no longer kept in registers of their own. The three
// instructions that consume them (is the key above greater? / take the new entry or the neighbour's / the smaller key) read lane-1
// through a DPP wave_shr:1 source operand (wave.h: wave_rank_fused; lane 0 has no source lane, is left alone by the hardware
// and so keeps the pre-set "new entry" / "new key"). gfx950 has no DPP form of v_cmp, so "q[lane-1] > qc" is v_max_i32_dpp +
// a plain compare: one instruction less per step than two maintained copies, measured 672 -> 636 ms on the slowest block of
// S-silesia, 316 -> 290 ms on a text block (A/B on the box: KNZ_RANK_VARIANT=4 is the kept-copies form). Measured on the
// same data (variants 7 / 8): four more instructions per low step cost 7 ns beside the chain and 5 ns on it: a lone wave
// pays per instruction issued, wherever it sits.
#ifndef KNZ_HIP_EMU
#include "rank_inv_asm.h"
#endif
template <int MODE, bool PACKED, bool WIDE = false, int XP = 0>
struct RankChainV {
    uint32_t xdummy = 0;                                                       // (XP: measurement variants, see step_low)
    uint32_t e[4], p[4];
    int q[4];
    uint32_t ep, pp, vff;
    int qp, lane;

    // (XP & 4, packed form: the lane-1 copies are not kept, the step takes them inside its instructions: wave_rank_fused)
    static constexpr bool FUSED = PACKED && (XP & 4) != 0;
    __device__ __forceinline__ void refresh() {
        if (FUSED) return;
        ep = wave_shr1(e[0]);
        if (!PACKED) pp = wave_shr1(p[0]);
        qp = (int)wave_shr1_keep0((uint32_t)qp, (uint32_t)q[0]);              // lane 0 keeps INT_MAX
    }
    __device__ __forceinline__ void init_identity(int l) {
        lane = l;
#pragma unroll
        for (int k = 0; k < 4; k++) { e[k] = 64u * (uint32_t)k + (uint32_t)l; p[k] = 0; q[k] = 0; }
        qp = 0x7FFFFFFF;
        vff = wave_in_vgpr(0xFFu);
        refresh();
    }
    static __device__ __forceinline__ int qf(uint32_t i, uint32_t pc) { return MODE == 1 ? (int)i : (int)((i + pc) >> 1); }

    // rank r <= 63; vi8 = (time << 8) and vi = time as wave-uniform VGPR values. Returns the entry (low byte = symbol).
    __device__ __forceinline__ uint32_t step_low(uint32_t r, uint32_t vi8, uint32_t vi) {
        const uint32_t se = wave_readlane(e[0], r);
        int vqc;
        uint32_t vnew;
#ifndef KNZ_HIP_EMU
        if (XP & 1) asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(xdummy) : "v"(vi8));   // 4 instructions beside the chain
        if (XP & 2) { uint32_t t; asm volatile("v_add_u32 %0, %1, %2\n\tv_sub_u32 %0, %0, %2\n\tv_add_u32 %0, %0, %2" : "=&v"(t) : "s"(se), "v"(vi8)); vi8 = t - se; }   // (X2: 4 instructions ON the chain)
#endif
        if (PACKED) { vqc = MODE == 1 ? (int)(vi8 >> 8) : (int)((se + vi8) >> 9); vnew = (se & vff) | vi8; }   // (i<<8) + (p<<8|sym) < 2^32 for i, p < 2^23
        else { vqc = MODE == 1 ? (int)vi : (int)((wave_readlane(p[0], r) + vi) >> 1); vnew = se; }
        const int qx = (int)wave_in_vgpr((uint32_t)lane > r ? 0x7FFFFFFFu : (uint32_t)q[0]);   // lanes above r always keep (opaque: or-ing two lane masks would go through the SALU)
        if (FUSED) {
            const bool keepf = qx > vqc;
            uint32_t t; int m;
            wave_rank_fused(q[0], e[0], vnew, vqc, t, m);
            e[0] = keepf ? e[0] : t;
            q[0] = keepf ? q[0] : m;
            return se;
        }
        const bool keep = qx > vqc, ins = qp > vqc;
        e[0] = keep ? e[0] : (ins ? vnew : ep);
        if (!PACKED) p[0] = keep ? p[0] : (ins ? vi : pp);
        q[0] = keep ? q[0] : min(vqc, qp);
        refresh();
        return se;
    }
    // `count` >= 1 consecutive ranks 0, the last one at time i (scalar): the symbol on top stays there
    __device__ __forceinline__ uint32_t run_top(uint32_t i, uint32_t count) {
        const uint32_t se = wave_readlane(e[0], 0);
        const uint32_t sym = PACKED ? (se & 0xFFu) : se;
        const uint32_t pc = count > 1 ? i - 1 : (PACKED ? se >> 8 : wave_readlane(p[0], 0));
        if (PACKED) e[0] = wave_writelane0(e[0], (i << 8) | sym); else p[0] = wave_writelane0(p[0], i);
        q[0] = (int)wave_writelane0((uint32_t)q[0], (uint32_t)qf(i, pc));
        refresh();
        return sym;
    }
    // rank 64..255 (high-entropy blocks: ~17 % of the ranks of S-silesia's executable-like member). The same select step, one
    // register of 64 ranks after the other from the one that holds r down to register 0; the lane-1 copy of a register takes its
    // lane 0 from lane 63 of the register below. Registers above r's are not touched.
    template <int K>
    __device__ __forceinline__ void high_reg(uint32_t r, int vqc, uint32_t vnew, uint32_t vi) {
        // the lane-1 copy of register K: lanes 1..63 from K itself, lane 0 from lane 63 of register K-1 (a rotate of K-1 puts it there:
        // two DPP moves, no v_readlane / v_writelane pair with its SGPR in between)
        uint32_t es, ps = 0u;
        int qs;
        if (K > 0) {
            es = wave_shr1_keep0(wave_ror1(e[K > 0 ? K - 1 : 0]), e[K]);
            if (!PACKED) ps = wave_shr1_keep0(wave_ror1(p[K > 0 ? K - 1 : 0]), p[K]);
            qs = (int)wave_shr1_keep0(wave_ror1((uint32_t)q[K > 0 ? K - 1 : 0]), (uint32_t)q[K]);
        } else if (FUSED) { es = wave_shr1(e[0]); qs = (int)wave_shr1_old((uint32_t)q[0], 0x7FFFFFFFu); }
        else { es = ep; ps = pp; qs = qp; }                                      // register 0: the maintained copies (q: lane 0 = +inf)
        const int qx = (int)wave_in_vgpr(64u * K + (uint32_t)lane > r ? 0x7FFFFFFFu : (uint32_t)q[K]);
        const bool keep = qx > vqc, ins = qs > vqc;
        e[K] = keep ? e[K] : (ins ? vnew : es);
        if (!PACKED) p[K] = keep ? p[K] : (ins ? vi : ps);
        q[K] = keep ? q[K] : min(vqc, qs);
    }
    template <int KR>
    __device__ __forceinline__ uint32_t step_high_in(uint32_t r, uint32_t vi8, uint32_t vi) {        // r lives in register KR
        const uint32_t l = r & 63;
        const uint32_t se = wave_readlane(e[KR], l);
        int vqc;
        uint32_t vnew;
        if (PACKED) { vqc = MODE == 1 ? (int)(vi8 >> 8) : (int)((se + vi8) >> 9); vnew = (se & vff) | vi8; }
        else { vqc = MODE == 1 ? (int)vi : (int)((wave_readlane(p[KR], l) + vi) >> 1); vnew = se; }
        if (KR >= 3) high_reg<3>(r, vqc, vnew, vi);
        if (KR >= 2) high_reg<2>(r, vqc, vnew, vi);
        high_reg<1>(r, vqc, vnew, vi);
        high_reg<0>(r, vqc, vnew, vi);
        refresh();
        return se;
    }
    __device__ __forceinline__ uint32_t step_high(uint32_t r, uint32_t vi8, uint32_t vi) {           // one straight-line body per register of r
        const uint32_t kr = r >> 6;
        if (kr == 1) return step_high_in<1>(r, vi8, vi);
        if (kr == 2) return step_high_in<2>(r, vi8, vi);
        return step_high_in<3>(r, vi8, vi);
    }
    // any rank, scalar time (tails, unaligned sources)
    __device__ __forceinline__ uint32_t step_any(uint32_t r, uint32_t i) {
        const uint32_t vi8 = wave_in_vgpr(i << 8), vi = wave_in_vgpr(i);
        return (r < 64 ? step_low(r, vi8, vi) : step_high(r, vi8, vi)) & (PACKED ? 0xFFu : 0xFFFFFFFFu);
    }
    // any rank below 64 * NR as ONE straight-line body: registers NR-1 .. 0 all take the select step (the ones above r's keep
    // everything: their lanes are "above r"), the accessed entry is read from the register that holds it through a select of the
    // NR registers (r >> 6 is known long before the entries are). No branch per symbol and nothing to reconcile behind one; the
    // steps of the extra registers run beside the critical path (a lone wave waits on latencies, not on issue slots).
    template <int NR>
    __device__ __forceinline__ uint32_t step_wide(uint32_t r, uint32_t vi8, uint32_t vi) {
        const uint32_t kr = r >> 6, l = r & 63;
        uint32_t esel = e[0], psel = p[0];
        if (NR > 1) { esel = kr == 1 ? e[1] : esel; if (!PACKED) psel = kr == 1 ? p[1] : psel; }
        if (NR > 2) { esel = kr == 2 ? e[2] : esel; if (!PACKED) psel = kr == 2 ? p[2] : psel; }
        if (NR > 3) { esel = kr == 3 ? e[3] : esel; if (!PACKED) psel = kr == 3 ? p[3] : psel; }
        const uint32_t se = wave_readlane(esel, l);
        int vqc;
        uint32_t vnew;
        if (PACKED) { vqc = MODE == 1 ? (int)(vi8 >> 8) : (int)((se + vi8) >> 9); vnew = (se & vff) | vi8; }
        else { vqc = MODE == 1 ? (int)vi : (int)((wave_readlane(psel, l) + vi) >> 1); vnew = se; }
        if (NR > 3) high_reg<3>(r, vqc, vnew, vi);
        if (NR > 2) high_reg<2>(r, vqc, vnew, vi);
        if (NR > 1) high_reg<1>(r, vqc, vnew, vi);
        high_reg<0>(r, vqc, vnew, vi);
        refresh();
        return se;
    }
    template <int W, int NR>
    __device__ __forceinline__ void word_wide(uint32_t w, uint32_t i, uint32_t& ob) {
        const uint32_t vi8 = wave_in_vgpr(i << 8), vi = wave_in_vgpr(i);
        ob = wave_writelane_c<4 * W>(ob, step_wide<NR>(w & 0xFFu, vi8, vi));
        ob = wave_writelane_c<4 * W + 1>(ob, step_wide<NR>((w >> 8) & 0xFFu, vi8 + 0x100u, vi + 1u));
        ob = wave_writelane_c<4 * W + 2>(ob, step_wide<NR>((w >> 16) & 0xFFu, vi8 + 0x200u, vi + 2u));
        ob = wave_writelane_c<4 * W + 3>(ob, step_wide<NR>(w >> 24, vi8 + 0x300u, vi + 3u));
    }
    // word W (0..3) of a group that holds ranks >= 64, first symbol at time i: one dispatch per symbol; symbols go to lanes 4W .. 4W+3 of ob
    template <int W>
    __device__ __forceinline__ void word_any(uint32_t w, uint32_t i, uint32_t& ob) {
        if ((w & 0xC0C0C0C0u) == 0) { word<W>(w, i, ob); return; }               // a clean word inside the group: no per-symbol dispatch
        if (w == 0) {
            const uint32_t s = run_top(i + 3, 4);
            ob = wave_writelane_c<4 * W>(ob, s); ob = wave_writelane_c<4 * W + 1>(ob, s);
            ob = wave_writelane_c<4 * W + 2>(ob, s); ob = wave_writelane_c<4 * W + 3>(ob, s);
            return;
        }
        if (WIDE) {                                                           // one branch per word: are all its ranks below 128?
            if ((w & 0x80808080u) == 0) word_wide<W, 2>(w, i, ob); else word_wide<W, 4>(w, i, ob);
            return;
        }
        const uint32_t vi8 = wave_in_vgpr(i << 8), vi = wave_in_vgpr(i);
        const uint32_t r0 = w & 0xFFu, r1 = (w >> 8) & 0xFFu, r2 = (w >> 16) & 0xFFu, r3 = w >> 24;
        // (the emulator and the measurement variants; the default device build takes groups with high ranks through rank_inv_asm.h. Measured: only
        // the low step as a hand-scheduled block between the compiler's branches is SLOWER, 617 -> 640 ms on the slowest block)
        ob = wave_writelane_c<4 * W>(ob, r0 < 64 ? step_low(r0, vi8, vi) : step_high(r0, vi8, vi));
        ob = wave_writelane_c<4 * W + 1>(ob, r1 < 64 ? step_low(r1, vi8 + 0x100u, vi + 1u) : step_high(r1, vi8 + 0x100u, vi + 1u));
        ob = wave_writelane_c<4 * W + 2>(ob, r2 < 64 ? step_low(r2, vi8 + 0x200u, vi + 2u) : step_high(r2, vi8 + 0x200u, vi + 2u));
        ob = wave_writelane_c<4 * W + 3>(ob, r3 < 64 ? step_low(r3, vi8 + 0x300u, vi + 3u) : step_high(r3, vi8 + 0x300u, vi + 3u));
    }
    // word W (0..3) of a 16-symbol group whose ranks are all < 64, first symbol at time i; symbols go to lanes 4W .. 4W+3 of
    // ob (low byte): no branch and no register traffic between the steps
    template <int W>
    __device__ __forceinline__ void word(uint32_t w, uint32_t i, uint32_t& ob) {
        if (w == 0) {
            const uint32_t s = run_top(i + 3, 4);
            ob = wave_writelane_c<4 * W>(ob, s); ob = wave_writelane_c<4 * W + 1>(ob, s);
            ob = wave_writelane_c<4 * W + 2>(ob, s); ob = wave_writelane_c<4 * W + 3>(ob, s);
            return;
        }
        const uint32_t vi8 = wave_in_vgpr(i << 8), vi = PACKED && MODE != 1 ? 0u : wave_in_vgpr(i);
        if (FUSED && MODE == 2) {                                                // the four steps as four hand-scheduled blocks
            const uint32_t vmaxi = wave_in_vgpr(0x7FFFFFFFu);
            wave_rank_step_packed<4 * W>(e[0], q[0], ob, w & 0xFFu, vi8, vff, (uint32_t)lane, vmaxi);
            wave_rank_step_packed<4 * W + 1>(e[0], q[0], ob, (w >> 8) & 0xFFu, vi8 + 0x100u, vff, (uint32_t)lane, vmaxi);
            wave_rank_step_packed<4 * W + 2>(e[0], q[0], ob, (w >> 16) & 0xFFu, vi8 + 0x200u, vff, (uint32_t)lane, vmaxi);
            wave_rank_step_packed<4 * W + 3>(e[0], q[0], ob, w >> 24, vi8 + 0x300u, vff, (uint32_t)lane, vmaxi);
            return;
        }
        ob = wave_writelane_c<4 * W>(ob, step_low(w & 0xFFu, vi8, vi));
        ob = wave_writelane_c<4 * W + 1>(ob, step_low((w >> 8) & 0xFFu, vi8 + 0x100u, vi + 1u));
        ob = wave_writelane_c<4 * W + 2>(ob, step_low((w >> 16) & 0xFFu, vi8 + 0x200u, vi + 2u));
        ob = wave_writelane_c<4 * W + 3>(ob, step_low(w >> 24, vi8 + 0x300u, vi + 3u));
    }
};

// ranks [begin, end) of a block through the chain `c`; begin is a multiple of 64 (whole output rows)
template <int MODE, bool PACKED, bool WIDE, int XP>
__device__ __forceinline__ void knz_rank_chain_range_v(RankChainV<MODE, PACKED, WIDE, XP>& c, const uint8_t* src, uint8_t* dst, uint32_t begin, uint32_t end, int lane) {
    uint32_t i0 = begin;
    if ((((uintptr_t)src) & 3) == 0) {                                          // (the pipeline's own regions are 16-byte aligned)
        // decoded symbols leave 64 at a time as one 64-byte row (round 3; 16 single bytes per group cost 1.5x the traffic: every line was
        // written in four partial pieces). Four groups are collected in one register, group q in byte q of lanes 0..15; a 4 x 4 byte
        // transpose inside every quad of lanes (two DPP moves + two byte permutes) turns that into four consecutive symbols per lane:
        // lane 4a + b then holds the dword 4b + a of the row.
        const bool rows = (((uintptr_t)dst) & 3) == 0;
        const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u, sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
        const uint32_t rowSlot = 4u * ((uint32_t)lane & 3u) + (((uint32_t)lane >> 2) & 3u);
        const uint32_t vmaxAll = wave_in_vgpr(0x7FFFFFFFu);                     // (loop invariant operand of the hand-written block)
        uint32_t racc = 0, rsh = 0;
        const uint32_t g0 = begin >> 4;
        const uint32_t gEnd = rows ? g0 + (((end - begin) >> 4) & ~3u) : g0;     // groups that lie in whole rows: the loop takes these, the last < 64 ranks go byte by byte below
#ifndef KNZ_HIP_EMU
        if (PACKED && MODE == 2 && (XP & 4) != 0 && !WIDE && !(XP & 3)) {           // default device form: the loop over the groups and the groups themselves hand-written (rank_inv_asm.h)
            if (gEnd > g0)
                knz_rank_rows_packed(c.e[0], c.e[1], c.e[2], c.e[3], c.q[0], c.q[1], c.q[2], c.q[3], src + 16 * (size_t)g0, 16u * (gEnd - g0), dst + 16 * (size_t)g0,
                                     4u * rowSlot, (16u * g0) << 8, c.vff, (uint32_t)lane, vmaxAll, sel1, sel2);
        } else
#endif
        {
        knz_u32x4 nxt = {0, 0, 0, 0};
        if (gEnd > g0) nxt = wave_sload_u32x4(src + 16 * (size_t)g0);
        for (uint32_t g = g0; g < gEnd; g++) {
            const knz_u32x4 cur = nxt;
            if (g + 1 < gEnd) nxt = wave_sload_u32x4(src + 16 * (size_t)(g + 1));  // one group ahead of the chain
            const uint32_t i = 16 * g;
            uint32_t ob = 0;
            const uint32_t any = cur.x | cur.y | cur.z | cur.w;
#ifndef KNZ_HIP_EMU
            if (PACKED && MODE == 2 && (XP & 4) != 0 && !WIDE) {                       // the group of sixteen accesses as one hand-written block (rank_inv_asm.h)
                knz_rank_group_packed(c.e[0], c.e[1], c.e[2], c.e[3], c.q[0], c.q[1], c.q[2], c.q[3], ob, cur.x, cur.y, cur.z, cur.w, i << 8, c.vff,
                                      (uint32_t)lane, vmaxAll);
            } else
#endif
            if (any == 0) ob = c.run_top(i + 15, 16);
            else if (any & 0xC0C0C0C0u) {                                        // ranks >= 64 in the group: words without one take the clean body
                c.template word_any<0>(cur.x, i, ob); c.template word_any<1>(cur.y, i + 4, ob);
                c.template word_any<2>(cur.z, i + 8, ob); c.template word_any<3>(cur.w, i + 12, ob);
            } else {
                c.template word<0>(cur.x, i, ob); c.template word<1>(cur.y, i + 4, ob);
                c.template word<2>(cur.z, i + 8, ob); c.template word<3>(cur.w, i + 12, ob);
            }
            racc |= (ob & 0xFFu) << rsh;
            rsh += 8;
            if (rsh == 32) {
                const uint32_t t = knz_byte_perm(wave_quad_xor1(racc), racc, sel1);
                const uint32_t o = knz_byte_perm(wave_quad_xor2(t), t, sel2);
                if (lane < 16) ((uint32_t*)(dst + (i - 48)))[rowSlot] = o;
                racc = 0; rsh = 0;
            }
        }
        }
        i0 = 16 * gEnd;
    }
    for (uint32_t i = i0; i < end; i++) {                                       // the last < 64 ranks (or an unaligned source): byte by byte
        const uint32_t s = c.step_any(wave_uniform(src[i]), i);
        if (lane == 0) dst[i] = (uint8_t)s;
    }
}

// The packed form (symbol and last access time in one register) holds while (i << 8) + (p << 8 | sym) fits 32 bits, i.e. for
// times below 2^23. An 8 MiB block behind a BWT is 2^23 + 25 ranks long (the primary indexes travel in front of the data,
// BWTBlockCodec.go:126-171): the chain runs packed up to time 2^23 and is unpacked there for the ranks that are left, instead
// of taking the three-register form for the whole block (what rounds 1-2 and the first half of round 3 did at BASELINE's block size).
#define KNZ_RANK_PACKED_TIMES (1u << 23)
template <int MODE, bool WIDE, int XP>
__device__ __forceinline__ void knz_rank_chain_block_v(const uint8_t* src, uint8_t* dst, uint32_t n, int lane, bool allowPacked, uint32_t cutRows) {
    if (!allowPacked) {
        RankChainV<MODE, false, WIDE, XP> c;
        c.init_identity(lane);
        knz_rank_chain_range_v<MODE, false, WIDE, XP>(c, src, dst, 0, n, lane);
        return;
    }
    RankChainV<MODE, true, WIDE, XP> c;
    c.init_identity(lane);
    const uint32_t cut = min(n, cutRows ? min(64u * cutRows, KNZ_RANK_PACKED_TIMES) : KNZ_RANK_PACKED_TIMES);   // (tests move the cut onto small inputs)
    knz_rank_chain_range_v<MODE, true, WIDE, XP>(c, src, dst, 0, cut, lane);
    if (cut == n) return;
    RankChainV<MODE, false, WIDE, XP> u;
    u.lane = lane;
#pragma unroll
    for (int k = 0; k < 4; k++) { u.e[k] = c.e[k] & 0xFFu; u.p[k] = c.e[k] >> 8; u.q[k] = c.q[k]; }
    u.qp = 0x7FFFFFFF;
    u.vff = c.vff;
    u.refresh();
    knz_rank_chain_range_v<MODE, false, WIDE, XP>(u, src, dst, cut, n, lane);
}

#if !defined(KNZ_HIP_EMU)
__device__ unsigned long long g_knz_rank_ticks[1024];                            // diagnostics (KNZ_RANK_PROF): 100 MHz ticks per block of the last launch
#define KNZ_RANK_NOW() (unsigned long long)__builtin_amdgcn_s_memrealtime()
#define KNZ_RANK_TICKS(b, t0) do { if (threadIdx.x == 0 && (b) < 1024u) g_knz_rank_ticks[b] = KNZ_RANK_NOW() - (t0); } while (0)
#else
#define KNZ_RANK_NOW() 0ull
#define KNZ_RANK_TICKS(b, t0) do { } while (0)
#endif
template <int MODE, int FLAGS>
__global__ __launch_bounds__(64) void knz_rank_inverse_chain_kernel(XfArgs a) {
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    const unsigned long long t0 = KNZ_RANK_NOW();
    if (!a.active[b]) { KNZ_RANK_TICKS(b, t0); return; }
    const uint32_t n = a.in_len[b];
    if (lane == 0) { a.out_len[b] = n; a.ok[b] = n <= a.out_cap ? 1 : -KNZ_ERR_PROCESS_BLOCK; }
    if (n > a.out_cap) return;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const bool packed = n <= (1u << 23) && !(a.mode & 0x100);                  // (bit 8 of mode: tests force the three-register form)
    if (FLAGS & 4) {
        knz_rank_chain_block_v<MODE, (FLAGS & 8) != 0, (FLAGS >> 4) & 7>(src, dst, n, lane, !(a.mode & 0x100), a.mode >> 12);
        KNZ_RANK_TICKS(b, t0);
    } else {
        if (packed) knz_rank_chain_block<MODE, true, FLAGS>(src, dst, n, lane);
        else knz_rank_chain_block<MODE, false, FLAGS>(src, dst, n, lane);
    }
}
