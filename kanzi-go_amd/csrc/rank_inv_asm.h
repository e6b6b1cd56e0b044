// Hand-written gfx950 blocks of the packed inverse RANK chain (rank_inv.hip: RankChainV<2, true>, device build only; the emulator
// runs the C++ form of the same steps). Why by hand: a lone wave pays ~2.4 ns for every instruction it issues, and the compiler's
// version of the per-symbol dispatch, unrolled sixteen times per group, reconciles the twelve live registers of the list behind
// every branch with v_mov_b32 (12 of the 27 instructions of a low step inside a word that holds a high rank) and pads hazards with
// s_nop. Here the list (e0..e3 = time << 8 | symbol, q0..q3 = key; ranks 64k + lane in register k) is updated IN PLACE, the
// branches are local to the block, and the instruction order needs no wait state:
//   * v_readlane (VALU writes an SGPR) -> two other instructions -> first VALU read of that SGPR;
//   * v_cmp_*_e64 to an SGPR pair -> at least two other instructions -> v_cndmask_b32_e64 reading the pair;
//   * a VGPR written by the VALU is read as a DPP source two or more instructions later (register k - 1 is read before it is
//     updated: the registers are walked from r's down).
// One access at rank r, time i (t8 = i << 8):
//   se = entry at rank r; qc = (se + t8) >> 9 = (i + p) >> 1; new = (se & 0xFF) | t8          [KNZ_RK_COMMON]
//   register k above r's: untouched; r's register: lanes above r & 63 keep                       [KNZ_RK_SELECT_PROT]
//   registers below: every lane is at or below r                                                 [KNZ_RK_SELECT, KNZ_RK_REG0]
//   per lane: keep if q > qc, else take `new` if q[lane-1] > qc, else the entry of lane-1; the lane-1 copy of register k > 0 has
//   its lane 0 from lane 63 of register k - 1 (wave_ror:1 of k - 1, then wave_shr:1 of k over it, which leaves lane 0 alone);
//   register 0 reads lane-1 through DPP operands and its lane 0 always takes the new entry.
#pragma once
#define KNZ_RK_DPP " row_mask:0xf bank_mask:0xf\n\t"
#define KNZ_RK_COMMON \
    "v_add_u32_e32 %[vqc], %[se], %[t8]\n\t" \
    "v_lshrrev_b32_e32 %[vqc], 9, %[vqc]\n\t" \
    "v_and_or_b32 %[vnew], %[se], %[vff], %[t8]\n\t"
#define KNZ_RK_COPIES(EK, QK, EL, QL) \
    "v_mov_b32_dpp %[es], %[" EL "] wave_ror:1" KNZ_RK_DPP \
    "v_mov_b32_dpp %[qs], %[" QL "] wave_ror:1" KNZ_RK_DPP \
    "v_mov_b32_dpp %[es], %[" EK "] wave_shr:1" KNZ_RK_DPP \
    "v_mov_b32_dpp %[qs], %[" QK "] wave_shr:1" KNZ_RK_DPP
#define KNZ_RK_SELECT_PROT(EK, QK) \
    "v_cmp_ge_u32_e32 vcc, %[l], %[lane]\n\t" \
    "v_cndmask_b32_e32 %[qx], %[vmax], %[" QK "], vcc\n\t" \
    "v_cmp_gt_i32_e64 %[keep], %[qx], %[vqc]\n\t" \
    "v_cmp_gt_i32_e32 vcc, %[qs], %[vqc]\n\t" \
    "v_cndmask_b32_e32 %[es], %[es], %[vnew], vcc\n\t" \
    "v_min_i32_e32 %[qs], %[vqc], %[qs]\n\t" \
    "v_cndmask_b32_e64 %[" EK "], %[es], %[" EK "], %[keep]\n\t" \
    "v_cndmask_b32_e64 %[" QK "], %[qs], %[" QK "], %[keep]\n\t"
#define KNZ_RK_SELECT(EK, QK) \
    "v_cmp_gt_i32_e64 %[keep], %[" QK "], %[vqc]\n\t" \
    "v_cmp_gt_i32_e32 vcc, %[qs], %[vqc]\n\t" \
    "v_cndmask_b32_e32 %[es], %[es], %[vnew], vcc\n\t" \
    "v_min_i32_e32 %[qs], %[vqc], %[qs]\n\t" \
    "v_cndmask_b32_e64 %[" EK "], %[es], %[" EK "], %[keep]\n\t" \
    "v_cndmask_b32_e64 %[" QK "], %[qs], %[" QK "], %[keep]\n\t"
#define KNZ_RK_REG0 \
    "v_cmp_gt_i32_e64 %[keep], %[q0], %[vqc]\n\t" \
    "v_max_i32_dpp %[qx], %[q0], %[vqc] wave_shr:1" KNZ_RK_DPP \
    "v_cmp_gt_i32_e32 vcc, %[qx], %[vqc]\n\t" \
    "v_cndmask_b32_dpp %[vnew], %[e0], %[vnew], vcc wave_shr:1" KNZ_RK_DPP \
    "v_min_i32_dpp %[vqc], %[q0], %[vqc] wave_shr:1" KNZ_RK_DPP \
    "v_cndmask_b32_e64 %[e0], %[vnew], %[e0], %[keep]\n\t" \
    "v_cndmask_b32_e64 %[q0], %[vqc], %[q0], %[keep]\n\t"
// rank %[r] < 64 at time %[t8]: 13 instructions, then the decoded entry to lane LN of ob
#define KNZ_RK_LOW(LN) \
    "v_readlane_b32 %[se], %[e0], %[r]\n\t" \
    "v_cmp_ge_u32_e32 vcc, %[r], %[lane]\n\t" \
    "v_cndmask_b32_e32 %[qx], %[vmax], %[q0], vcc\n\t" \
    KNZ_RK_COMMON \
    "v_cmp_gt_i32_e64 %[keep], %[qx], %[vqc]\n\t" \
    "v_max_i32_dpp %[qx], %[q0], %[vqc] wave_shr:1" KNZ_RK_DPP \
    "v_cmp_gt_i32_e32 vcc, %[qx], %[vqc]\n\t" \
    "v_cndmask_b32_dpp %[vnew], %[e0], %[vnew], vcc wave_shr:1" KNZ_RK_DPP \
    "v_min_i32_dpp %[vqc], %[q0], %[vqc] wave_shr:1" KNZ_RK_DPP \
    "v_cndmask_b32_e64 %[e0], %[vnew], %[e0], %[keep]\n\t" \
    "v_cndmask_b32_e64 %[q0], %[vqc], %[q0], %[keep]\n"
// symbol (W, N) of a word that holds a high rank: EXTRACT puts its rank into %[r], TIME its time << 8 into %[t8]; a rank below 64
// falls through, the others go out of line (KNZ_RK_HIGH) and come back to the write of the output lane
#define KNZ_RK_SYMBOL(ID, EXTRACT, TIME, LN) \
    EXTRACT TIME \
    "s_cmp_gt_u32 %[r], 63\n\t" \
    "s_cbranch_scc1 .Lknz_rk_high" ID "_%=\n\t" \
    KNZ_RK_LOW(LN) \
    ".Lknz_rk_ret" ID "_%=:\n\t" \
    "v_writelane_b32 %[ob], %[se], " LN "\n\t"
#define KNZ_RK_HIGH(ID) \
    ".Lknz_rk_high" ID "_%=:\n\t" \
    "s_and_b32 %[l], %[r], 63\n\t" \
    "s_cmp_lt_u32 %[r], 128\n\t" \
    "s_cbranch_scc1 .Lknz_rk_k1_" ID "_%=\n\t" \
    "s_cmp_lt_u32 %[r], 192\n\t" \
    "s_cbranch_scc1 .Lknz_rk_k2_" ID "_%=\n\t" \
    "v_readlane_b32 %[se], %[e3], %[l]\n\t" \
    KNZ_RK_COPIES("e3", "q3", "e2", "q2") \
    KNZ_RK_COMMON \
    KNZ_RK_SELECT_PROT("e3", "q3") \
    KNZ_RK_COPIES("e2", "q2", "e1", "q1") \
    KNZ_RK_SELECT("e2", "q2") \
    KNZ_RK_COPIES("e1", "q1", "e0", "q0") \
    KNZ_RK_SELECT("e1", "q1") \
    KNZ_RK_REG0 \
    "s_branch .Lknz_rk_ret" ID "_%=\n" \
    ".Lknz_rk_k2_" ID "_%=:\n\t" \
    "v_readlane_b32 %[se], %[e2], %[l]\n\t" \
    KNZ_RK_COPIES("e2", "q2", "e1", "q1") \
    KNZ_RK_COMMON \
    KNZ_RK_SELECT_PROT("e2", "q2") \
    KNZ_RK_COPIES("e1", "q1", "e0", "q0") \
    KNZ_RK_SELECT("e1", "q1") \
    KNZ_RK_REG0 \
    "s_branch .Lknz_rk_ret" ID "_%=\n" \
    ".Lknz_rk_k1_" ID "_%=:\n\t" \
    "v_readlane_b32 %[se], %[e1], %[l]\n\t" \
    KNZ_RK_COPIES("e1", "q1", "e0", "q0") \
    KNZ_RK_COMMON \
    KNZ_RK_SELECT_PROT("e1", "q1") \
    KNZ_RK_REG0 \
    "s_branch .Lknz_rk_ret" ID "_%=\n"
#define KNZ_RK_X0(WR) "s_and_b32 %[r], %[" WR "], 0xff\n\t"
#define KNZ_RK_X1(WR) "s_bfe_u32 %[r], %[" WR "], 0x80008\n\t"
#define KNZ_RK_X2(WR) "s_bfe_u32 %[r], %[" WR "], 0x80010\n\t"
#define KNZ_RK_X3(WR) "s_lshr_b32 %[r], %[" WR "], 24\n\t"
#define KNZ_RK_T(K) "v_add_u32_e32 %[t8], " K ", %[vbase]\n\t"
// word W of a group that holds a high rank (%[wW] = its four ranks, T0..T3 = (4W + n) << 8, L0..L3 = 4W + n):
//   a high rank in the word: four dispatched symbols, inline; otherwise out of line (KNZ_RK_WORD_REST): four ranks 0 = the entry on
//   top stays there (lane 0 of e0 / q0 rewritten once), or four low steps without a dispatch
#define KNZ_RK_WORD(W, WR, T0, T1, T2, T3, L0, L1, L2, L3) \
    "s_and_b32 %[r], %[" WR "], 0xc0c0c0c0\n\t" \
    "s_cmp_eq_u32 %[r], 0\n\t" \
    "s_cbranch_scc1 .Lknz_rk_rest" W "_%=\n\t" \
    KNZ_RK_SYMBOL(W "0", KNZ_RK_X0(WR), KNZ_RK_T(T0), L0) \
    KNZ_RK_SYMBOL(W "1", KNZ_RK_X1(WR), KNZ_RK_T(T1), L1) \
    KNZ_RK_SYMBOL(W "2", KNZ_RK_X2(WR), KNZ_RK_T(T2), L2) \
    KNZ_RK_SYMBOL(W "3", KNZ_RK_X3(WR), KNZ_RK_T(T3), L3) \
    ".Lknz_rk_wend" W "_%=:\n\t"
#define KNZ_RK_WORD_REST(W, WR, T0, T1, T2, T3, Q3, L0, L1, L2, L3) \
    ".Lknz_rk_rest" W "_%=:\n\t" \
    "s_cmp_eq_u32 %[" WR "], 0\n\t" \
    "s_cbranch_scc1 .Lknz_rk_zero" W "_%=\n\t" \
    KNZ_RK_X0(WR) KNZ_RK_T(T0) KNZ_RK_LOW(L0) "\tv_writelane_b32 %[ob], %[se], " L0 "\n\t" \
    KNZ_RK_X1(WR) KNZ_RK_T(T1) KNZ_RK_LOW(L1) "\tv_writelane_b32 %[ob], %[se], " L1 "\n\t" \
    KNZ_RK_X2(WR) KNZ_RK_T(T2) KNZ_RK_LOW(L2) "\tv_writelane_b32 %[ob], %[se], " L2 "\n\t" \
    KNZ_RK_X3(WR) KNZ_RK_T(T3) KNZ_RK_LOW(L3) "\tv_writelane_b32 %[ob], %[se], " L3 "\n\t" \
    "s_branch .Lknz_rk_wend" W "_%=\n" \
    ".Lknz_rk_zero" W "_%=:\n\t" \
    "v_readlane_b32 %[se], %[e0], 0\n\t"                     /* four ranks 0 at times i+4W .. i+4W+3 */ \
    "s_and_b32 %[se], %[se], 0xff\n\t" \
    "s_add_i32 %[l], %[i8], " T3 "\n\t"                      /* e0[0] = (last time << 8) | symbol */ \
    "s_or_b32 %[l], %[l], %[se]\n\t" \
    "v_writelane_b32 %[e0], %[l], 0\n\t" \
    "s_lshr_b32 %[l], %[i8], 8\n\t"                          /* q0[0] = ((i+4W+3) + (i+4W+2)) >> 1 = i + 4W + 2 */ \
    "s_add_i32 %[l], %[l], " Q3 "\n\t" \
    "v_writelane_b32 %[q0], %[l], 0\n\t" \
    "v_writelane_b32 %[ob], %[se], " L0 "\n\t" \
    "v_writelane_b32 %[ob], %[se], " L1 "\n\t" \
    "v_writelane_b32 %[ob], %[se], " L2 "\n\t" \
    "v_writelane_b32 %[ob], %[se], " L3 "\n\t" \
    "s_branch .Lknz_rk_wend" W "_%=\n"
// ---- a group WITHOUT a rank of 64 or more (every group of a text-like block): words of four ranks 0 go out of line, the others are
// four low steps in a row; sixteen ranks 0 are one step
#define KNZ_RK_ZERO4(W, T3, Q3, L0, L1, L2, L3) \
    ".Lknz_rk_bzero" W "_%=:\n\t" \
    "v_readlane_b32 %[se], %[e0], 0\n\t" \
    "s_and_b32 %[se], %[se], 0xff\n\t" \
    "s_add_i32 %[l], %[i8], " T3 "\n\t" \
    "s_or_b32 %[l], %[l], %[se]\n\t" \
    "v_writelane_b32 %[e0], %[l], 0\n\t" \
    "s_lshr_b32 %[l], %[i8], 8\n\t" \
    "s_add_i32 %[l], %[l], " Q3 "\n\t" \
    "v_writelane_b32 %[q0], %[l], 0\n\t" \
    "v_writelane_b32 %[ob], %[se], " L0 "\n\t" \
    "v_writelane_b32 %[ob], %[se], " L1 "\n\t" \
    "v_writelane_b32 %[ob], %[se], " L2 "\n\t" \
    "v_writelane_b32 %[ob], %[se], " L3 "\n\t" \
    "s_branch .Lknz_rk_bwend" W "_%=\n"
#define KNZ_RK_CWORD(W, WR, T0, T1, T2, T3, L0, L1, L2, L3) \
    "s_cmp_eq_u32 %[" WR "], 0\n\t" \
    "s_cbranch_scc1 .Lknz_rk_bzero" W "_%=\n\t" \
    KNZ_RK_X0(WR) KNZ_RK_T(T0) KNZ_RK_LOW(L0) "\tv_writelane_b32 %[ob], %[se], " L0 "\n\t" \
    KNZ_RK_X1(WR) KNZ_RK_T(T1) KNZ_RK_LOW(L1) "\tv_writelane_b32 %[ob], %[se], " L1 "\n\t" \
    KNZ_RK_X2(WR) KNZ_RK_T(T2) KNZ_RK_LOW(L2) "\tv_writelane_b32 %[ob], %[se], " L2 "\n\t" \
    KNZ_RK_X3(WR) KNZ_RK_T(T3) KNZ_RK_LOW(L3) "\tv_writelane_b32 %[ob], %[se], " L3 "\n" \
    ".Lknz_rk_bwend" W "_%=:\n\t"

// ---- sixteen accesses (the ranks in w0..w3, first one at time i; i8 = i << 8 as a scalar); decoded entries to lanes 0..15 of ob.
// Both kinds of group behind ONE statement: with a statement per kind the compiler reconciles the list's registers where their paths
// meet and again on the loop's back edge (30-40 v_mov_b32 per group: measured 460 -> 418 ms on the slowest block)
__device__ __forceinline__ void knz_rank_group_packed(uint32_t& e0, uint32_t& e1, uint32_t& e2, uint32_t& e3, int& q0, int& q1, int& q2, int& q3, uint32_t& ob,
                                                      uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t i8, uint32_t vff, uint32_t lane, uint32_t vmax) {
    uint32_t se, l, r, vnew, es, t8, vbase;
    int qx, vqc, qs;
    uint64_t keep;
    asm volatile(
        "s_or_b32 %[r], %[w0], %[w1]\n\t"
        "s_or_b32 %[l], %[w2], %[w3]\n\t"
        "s_or_b32 %[r], %[r], %[l]\n\t"
        "v_mov_b32_e32 %[vbase], %[i8]\n\t"
        "s_and_b32 %[l], %[r], 0xc0c0c0c0\n\t"
        "s_cmp_eq_u32 %[l], 0\n\t"
        "s_cbranch_scc1 .Lknz_rk_lowgroup_%=\n\t"
        // a group that holds a rank of 64 or more
        KNZ_RK_WORD("0", "w0", "0", "0x100", "0x200", "0x300", "0", "1", "2", "3")
        KNZ_RK_WORD("1", "w1", "0x400", "0x500", "0x600", "0x700", "4", "5", "6", "7")
        KNZ_RK_WORD("2", "w2", "0x800", "0x900", "0xa00", "0xb00", "8", "9", "10", "11")
        KNZ_RK_WORD("3", "w3", "0xc00", "0xd00", "0xe00", "0xf00", "12", "13", "14", "15")
        "s_branch .Lknz_rk_done_%=\n"
        KNZ_RK_WORD_REST("0", "w0", "0", "0x100", "0x200", "0x300", "2", "0", "1", "2", "3")
        KNZ_RK_WORD_REST("1", "w1", "0x400", "0x500", "0x600", "0x700", "6", "4", "5", "6", "7")
        KNZ_RK_WORD_REST("2", "w2", "0x800", "0x900", "0xa00", "0xb00", "10", "8", "9", "10", "11")
        KNZ_RK_WORD_REST("3", "w3", "0xc00", "0xd00", "0xe00", "0xf00", "14", "12", "13", "14", "15")
        KNZ_RK_HIGH("00") KNZ_RK_HIGH("01") KNZ_RK_HIGH("02") KNZ_RK_HIGH("03")
        KNZ_RK_HIGH("10") KNZ_RK_HIGH("11") KNZ_RK_HIGH("12") KNZ_RK_HIGH("13")
        KNZ_RK_HIGH("20") KNZ_RK_HIGH("21") KNZ_RK_HIGH("22") KNZ_RK_HIGH("23")
        KNZ_RK_HIGH("30") KNZ_RK_HIGH("31") KNZ_RK_HIGH("32") KNZ_RK_HIGH("33")
        // a group without one (%[r] = the OR of its sixteen ranks)
        ".Lknz_rk_lowgroup_%=:\n\t"
        "s_cmp_eq_u32 %[r], 0\n\t"
        "s_cbranch_scc1 .Lknz_rk_zero16_%=\n\t"
        KNZ_RK_CWORD("0", "w0", "0", "0x100", "0x200", "0x300", "0", "1", "2", "3")
        KNZ_RK_CWORD("1", "w1", "0x400", "0x500", "0x600", "0x700", "4", "5", "6", "7")
        KNZ_RK_CWORD("2", "w2", "0x800", "0x900", "0xa00", "0xb00", "8", "9", "10", "11")
        KNZ_RK_CWORD("3", "w3", "0xc00", "0xd00", "0xe00", "0xf00", "12", "13", "14", "15")
        "s_branch .Lknz_rk_done_%=\n"
        KNZ_RK_ZERO4("0", "0x300", "2", "0", "1", "2", "3")
        KNZ_RK_ZERO4("1", "0x700", "6", "4", "5", "6", "7")
        KNZ_RK_ZERO4("2", "0xb00", "10", "8", "9", "10", "11")
        KNZ_RK_ZERO4("3", "0xf00", "14", "12", "13", "14", "15")
        ".Lknz_rk_zero16_%=:\n\t"
        "v_readlane_b32 %[se], %[e0], 0\n\t"
        "s_and_b32 %[se], %[se], 0xff\n\t"
        "s_add_i32 %[l], %[i8], 0xf00\n\t"
        "s_or_b32 %[l], %[l], %[se]\n\t"
        "v_writelane_b32 %[e0], %[l], 0\n\t"
        "s_lshr_b32 %[l], %[i8], 8\n\t"
        "s_add_i32 %[l], %[l], 14\n\t"
        "v_writelane_b32 %[q0], %[l], 0\n\t"
        "v_mov_b32_e32 %[ob], %[se]\n"
        ".Lknz_rk_done_%=:"
        : [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2), [e3] "+v"(e3), [q0] "+v"(q0), [q1] "+v"(q1), [q2] "+v"(q2), [q3] "+v"(q3), [ob] "+v"(ob),
          [se] "=&s"(se), [l] "=&s"(l), [r] "=&s"(r), [vnew] "=&v"(vnew), [es] "=&v"(es), [qx] "=&v"(qx), [vqc] "=&v"(vqc), [qs] "=&v"(qs), [t8] "=&v"(t8),
          [vbase] "=&v"(vbase), [keep] "=&s"(keep)
        : [w0] "s"(w0), [w1] "s"(w1), [w2] "s"(w2), [w3] "s"(w3), [i8] "s"(i8), [vff] "v"(vff), [lane] "v"(lane), [vmax] "v"(vmax)
        : "vcc", "scc");
}

// ---- the loop around the block as well: `nbytes` / 16 groups of sixteen ranks starting at src (nbytes a multiple of 64), the decoded bytes
// to dbase in rows of 64 (lane l < 16 stores the dword at doff = 4 * rowSlot(l), see rank_inv.hip, + 64 per row). The ranks of a group arrive
// through one s_load_dwordx4 issued a group ahead (into s92..s95, moved to s88..s91 = w0..w3 when the group starts: the registers are named
// because the halves of a loaded quad are operands); four groups are collected in `racc` (byte k of lane j = symbol j of group k) and leave
// through the 4 x 4 byte transpose inside every quad of lanes. Per group the loop costs ~16 instructions; the compiler's loop around the
// one-group statement cost ~30 (measured: slowest block 418 -> 408 ms, profiles/r03_rank_inverse_per_block.txt).
__device__ __forceinline__ void knz_rank_rows_packed(uint32_t& e0, uint32_t& e1, uint32_t& e2, uint32_t& e3, int& q0, int& q1, int& q2, int& q3,
                                                     const uint8_t* src, uint32_t nbytes, uint8_t* dbase, uint32_t doff, uint32_t i8,
                                                     uint32_t vff, uint32_t lane, uint32_t vmax, uint32_t sel1, uint32_t sel2) {
    uint32_t se, l, lo, r, vnew, es, t8, vbase, ob, racc, tt, soff, rsh, w0, w1, w2, w3;
    int qx, vqc, qs;
    uint64_t keep;
    const uint32_t lastoff = nbytes - 16;
    asm volatile(
        "s_load_dwordx4 s[88:91], %[src], 0x0\n\t"
        "s_mov_b32 %[soff], 16\n\t"
        "s_mov_b32 %[rsh], 0\n\t"
        "v_mov_b32_e32 %[vbase], %[i8]\n\t"
        "v_mov_b32_e32 %[racc], 0\n\t"
        "s_waitcnt lgkmcnt(0)\n"
        ".Lknz_rk_loop_%=:\n\t"
        "s_min_u32 %[lo], %[soff], %[lastoff]\n\t"               /* the group after this one (the last group is read twice rather than reading past the end) */
        "s_load_dwordx4 s[92:95], %[src], %[lo]\n\t"             /* (%[lo] stays untouched until the wait at the bottom of the loop) */
        "s_or_b32 %[r], %[w0], %[w1]\n\t"
        "s_or_b32 %[l], %[w2], %[w3]\n\t"
        "s_or_b32 %[r], %[r], %[l]\n\t"
        "s_and_b32 %[l], %[r], 0xc0c0c0c0\n\t"
        "s_cmp_eq_u32 %[l], 0\n\t"
        "s_cbranch_scc1 .Lknz_rk_lowgroup_%=\n\t"
        KNZ_RK_WORD("0", "w0", "0", "0x100", "0x200", "0x300", "0", "1", "2", "3")
        KNZ_RK_WORD("1", "w1", "0x400", "0x500", "0x600", "0x700", "4", "5", "6", "7")
        KNZ_RK_WORD("2", "w2", "0x800", "0x900", "0xa00", "0xb00", "8", "9", "10", "11")
        KNZ_RK_WORD("3", "w3", "0xc00", "0xd00", "0xe00", "0xf00", "12", "13", "14", "15")
        ".Lknz_rk_done_%=:\n\t"
        "v_and_b32_e32 %[tt], 0xff, %[ob]\n\t"
        "v_lshl_or_b32 %[racc], %[tt], %[rsh], %[racc]\n\t"
        "v_add_u32_e32 %[vbase], 0x1000, %[vbase]\n\t"
        "s_addk_i32 %[i8], 0x1000\n\t"
        "s_add_u32 %[soff], %[soff], 16\n\t"
        "s_add_u32 %[rsh], %[rsh], 8\n\t"
        "s_cmp_lg_u32 %[rsh], 32\n\t"
        "s_cbranch_scc1 .Lknz_rk_norow_%=\n\t"
        "v_mov_b32_dpp %[tt], %[racc] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_perm_b32 %[tt], %[tt], %[racc], %[sel1]\n\t"
        "s_mov_b32 %[rsh], 0\n\t"
        "s_nop 0\n\t"
        "v_mov_b32_dpp %[racc], %[tt] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_perm_b32 %[tt], %[racc], %[tt], %[sel2]\n\t"
        "s_mov_b64 exec, 0xffff\n\t"
        "global_store_dword %[doff], %[tt], %[dbase]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "v_add_u32_e32 %[doff], 64, %[doff]\n\t"
        "v_mov_b32_e32 %[racc], 0\n"
        ".Lknz_rk_norow_%=:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b64 s[88:89], s[92:93]\n\t"
        "s_mov_b64 s[90:91], s[94:95]\n\t"
        "s_cmp_le_u32 %[soff], %[nbytes]\n\t"                   /* soff = 16 x (groups done + 1) */
        "s_cbranch_scc1 .Lknz_rk_loop_%=\n\t"
        "s_branch .Lknz_rk_end_%=\n"
        KNZ_RK_WORD_REST("0", "w0", "0", "0x100", "0x200", "0x300", "2", "0", "1", "2", "3")
        KNZ_RK_WORD_REST("1", "w1", "0x400", "0x500", "0x600", "0x700", "6", "4", "5", "6", "7")
        KNZ_RK_WORD_REST("2", "w2", "0x800", "0x900", "0xa00", "0xb00", "10", "8", "9", "10", "11")
        KNZ_RK_WORD_REST("3", "w3", "0xc00", "0xd00", "0xe00", "0xf00", "14", "12", "13", "14", "15")
        KNZ_RK_HIGH("00") KNZ_RK_HIGH("01") KNZ_RK_HIGH("02") KNZ_RK_HIGH("03")
        KNZ_RK_HIGH("10") KNZ_RK_HIGH("11") KNZ_RK_HIGH("12") KNZ_RK_HIGH("13")
        KNZ_RK_HIGH("20") KNZ_RK_HIGH("21") KNZ_RK_HIGH("22") KNZ_RK_HIGH("23")
        KNZ_RK_HIGH("30") KNZ_RK_HIGH("31") KNZ_RK_HIGH("32") KNZ_RK_HIGH("33")
        ".Lknz_rk_lowgroup_%=:\n\t"
        "s_cmp_eq_u32 %[r], 0\n\t"
        "s_cbranch_scc1 .Lknz_rk_zero16_%=\n\t"
        KNZ_RK_CWORD("0", "w0", "0", "0x100", "0x200", "0x300", "0", "1", "2", "3")
        KNZ_RK_CWORD("1", "w1", "0x400", "0x500", "0x600", "0x700", "4", "5", "6", "7")
        KNZ_RK_CWORD("2", "w2", "0x800", "0x900", "0xa00", "0xb00", "8", "9", "10", "11")
        KNZ_RK_CWORD("3", "w3", "0xc00", "0xd00", "0xe00", "0xf00", "12", "13", "14", "15")
        "s_branch .Lknz_rk_done_%=\n"
        KNZ_RK_ZERO4("0", "0x300", "2", "0", "1", "2", "3")
        KNZ_RK_ZERO4("1", "0x700", "6", "4", "5", "6", "7")
        KNZ_RK_ZERO4("2", "0xb00", "10", "8", "9", "10", "11")
        KNZ_RK_ZERO4("3", "0xf00", "14", "12", "13", "14", "15")
        ".Lknz_rk_zero16_%=:\n\t"
        "v_readlane_b32 %[se], %[e0], 0\n\t"
        "s_and_b32 %[se], %[se], 0xff\n\t"
        "s_add_i32 %[l], %[i8], 0xf00\n\t"
        "s_or_b32 %[l], %[l], %[se]\n\t"
        "v_writelane_b32 %[e0], %[l], 0\n\t"
        "s_lshr_b32 %[l], %[i8], 8\n\t"
        "s_add_i32 %[l], %[l], 14\n\t"
        "v_writelane_b32 %[q0], %[l], 0\n\t"
        "v_mov_b32_e32 %[ob], %[se]\n\t"
        "s_branch .Lknz_rk_done_%=\n"
        ".Lknz_rk_end_%=:"
        : [e0] "+v"(e0), [e1] "+v"(e1), [e2] "+v"(e2), [e3] "+v"(e3), [q0] "+v"(q0), [q1] "+v"(q1), [q2] "+v"(q2), [q3] "+v"(q3),
          [doff] "+v"(doff), [i8] "+s"(i8),
          [se] "=&s"(se), [l] "=&s"(l), [r] "=&s"(r), [lo] "=&s"(lo), [soff] "=&s"(soff), [rsh] "=&s"(rsh), [keep] "=&s"(keep),
          [w0] "=&{s88}"(w0), [w1] "=&{s89}"(w1), [w2] "=&{s90}"(w2), [w3] "=&{s91}"(w3),
          [vnew] "=&v"(vnew), [es] "=&v"(es), [qx] "=&v"(qx), [vqc] "=&v"(vqc), [qs] "=&v"(qs), [t8] "=&v"(t8), [vbase] "=&v"(vbase),
          [ob] "=&v"(ob), [racc] "=&v"(racc), [tt] "=&v"(tt)
        : [src] "s"(src), [dbase] "s"(dbase), [nbytes] "s"(nbytes), [lastoff] "s"(lastoff), [vff] "v"(vff), [lane] "v"(lane), [vmax] "v"(vmax),
          [sel1] "v"(sel1), [sel2] "v"(sel2)
        : "vcc", "scc", "s92", "s93", "s94", "s95", "memory");
}
