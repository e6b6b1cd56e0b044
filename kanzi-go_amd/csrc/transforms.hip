// Byte transforms of the hot path on gfx950: SBRT (RANK / MTFT) and ZRLT, forward and inverse, batched over the blocks
// of a stream. Replaces SBRT.Forward/Inverse (v2/transform/SBRT.go:127-175,180-226) and ZRLT.Forward/Inverse
// (v2/transform/ZRLT.go:58-137,142-225) inside ByteTransformSequence (v2/transform/Sequence.go:64-186).
//
// Forward transforms are segment-parallel:
//   RANK/MTFT: the list-update state at any position only depends on the last two accesses of every symbol, so every
//              8 KiB segment finds its symbols' last two positions (parallel), a per-block pass carries them across
//              segments, and then all segments replay the list update independently from the reconstructed state.
//   ZRLT:      element sizes (literal 1|2 bytes, zero run of k -> floor(log2(k+1)) bytes at the run's last zero) are
//              scanned per segment and per block, then scattered; the "must not expand" rule becomes a test on the total.
// Inverse transforms are serial by construction of the format (one chain per block), staged through LDS tiles so that
// global memory is only touched with coalesced accesses.
#include "bits.h"

#define KNZ_SEG 8192

struct XfArgs {
    uint32_t nblocks;
    uint32_t segs_per_block;
    const uint64_t* in_ptr;       // [nblocks] device address of the block's current bytes
    const uint32_t* in_len;       // [nblocks]
    const uint64_t* out_ptr;      // [nblocks] where this stage writes
    uint32_t out_cap;             // bytes available per block at out_ptr
    uint32_t* out_len;            // [nblocks]
    int32_t* ok;                  // [nblocks] 1 applied, 0 declined (transform skipped), <0: -(kanzi error)
    const uint8_t* active;        // [nblocks] 0 = block does not run this stage (copy block / earlier failure)
    int32_t* seg_a;               // per-segment scratch (meaning depends on the transform)
    int32_t* seg_b;
    uint32_t mode;                // SBRT mode: 1 MTF, 2 RANK, 3 TIMESTAMP
};

// After a stage: applied blocks switch to the stage output and clear their skip bit (Sequence.go:93-95)
struct CommitArgs {
    uint32_t nblocks; uint32_t stage;
    const int32_t* ok; const uint32_t* out_len; const uint64_t* out_ptr;
    uint64_t* cur_ptr; uint32_t* cur_len; uint8_t* skip; uint8_t* side; uint8_t target_if_side1, target_else;
    uint64_t base1, base2, stride; int32_t* blk_status;
};
__global__ void knz_xf_commit_kernel(CommitArgs a) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks) return;
    const int32_t ok = a.ok[b];
    if (ok == 1) {
        a.cur_ptr[b] = a.out_ptr[b];
        a.cur_len[b] = a.out_len[b];
        a.skip[b] &= (uint8_t)~(1u << (7 - a.stage));
        a.side[b] = a.side[b] == 1 ? 2 : 1;
    } else if (ok < 0) {
        a.blk_status[b] = -ok;
    }
}
// Before a stage: out_ptr = the region the block is not currently in
__global__ void knz_xf_prepare_kernel(CommitArgs a, uint64_t* out_ptr, int32_t* ok) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks) return;
    out_ptr[b] = (a.side[b] == 1 ? a.base2 : a.base1) + (uint64_t)b * a.stride;
    ok[b] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// SBRT forward
__device__ __forceinline__ int knz_sbrt_q(uint32_t mode, int i, int p) {
    // qc = ((i & mask1) + (p[c] & mask2)) >> shift   (SBRT.go:59-76,158)
    const int m1 = mode == 3 ? 0 : -1, m2 = mode == 1 ? 0 : -1, s = mode == 2 ? 1 : 0;
    return ((i & m1) + (p & m2)) >> s;
}

// 1) last two positions of every symbol inside the segment (block-local positions, -1 if none)
__global__ __launch_bounds__(64) void knz_sbrt_seg_last2_kernel(XfArgs a) {
    __shared__ int s_last[256], s_prev[256];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    int32_t* oa = a.seg_a + (size_t)blockIdx.x * 256;
    int32_t* ob = a.seg_b + (size_t)blockIdx.x * 256;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG);
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    for (int i = lane; i < 256; i += 64) { s_last[i] = -1; s_prev[i] = -1; }
    wave_sync();
    for (uint32_t i = lo + lane; i < hi; i += 64) atomicMax(&s_last[src[i]], (int)i);
    wave_sync();
    for (uint32_t i = lo + lane; i < hi; i += 64) { const uint8_t c = src[i]; if ((int)i != s_last[c]) atomicMax(&s_prev[c], (int)i); }
    wave_sync();
    for (int i = lane; i < 256; i += 64) { oa[i] = s_last[i]; ob[i] = s_prev[i]; }
}

// 2) carry across the segments of a block: on exit seg_a/seg_b hold (last, prev) access of each symbol BEFORE the segment
__global__ __launch_bounds__(256) void knz_sbrt_carry_kernel(XfArgs a) {
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const int d = threadIdx.x;
    const uint32_t n = a.in_len[b];
    const uint32_t nseg = (n + KNZ_SEG - 1) / KNZ_SEG;
    int last = -1, prev = -1;
    for (uint32_t s = 0; s < nseg; s++) {
        int32_t* pa = a.seg_a + ((size_t)b * a.segs_per_block + s) * 256 + d;
        int32_t* pb = a.seg_b + ((size_t)b * a.segs_per_block + s) * 256 + d;
        const int li = *pa, pi = *pb;
        *pa = last; *pb = prev;
        if (li >= 0) { prev = pi >= 0 ? pi : last; last = li; }
    }
}

// 3) replay of the list update inside each segment from the reconstructed state (SBRT.go:155-172)
__global__ __launch_bounds__(64) void knz_sbrt_apply_kernel(XfArgs a) {
    __shared__ uint8_t s_in[KNZ_SEG], s_out[KNZ_SEG];
    __shared__ uint8_t s_s2r[256], s_r2s[256];
    __shared__ int s_p[256], s_q[256], s_t[256];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    if (s == 0 && lane == 0) { a.out_len[b] = n; a.ok[b] = (a.out_cap >= n + 33) ? 1 : 0; }   // MaxEncodedLen check (:137-139)
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG), cnt = hi - lo;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const int32_t* ca = a.seg_a + (size_t)blockIdx.x * 256;
    const int32_t* cb = a.seg_b + (size_t)blockIdx.x * 256;
    for (uint32_t i = lane; i < cnt; i += 64) s_in[i] = src[lo + i];
    for (int d = lane; d < 256; d += 64) {
        const int last = ca[d], prev = cb[d];
        s_t[d] = last;                                            // -1 = never accessed
        s_p[d] = last >= 0 ? last : 0;                            // p[] starts at 0 (:150)
        s_q[d] = last >= 0 ? knz_sbrt_q(a.mode, last, prev >= 0 ? prev : 0) : 0;
    }
    wave_sync();
    // list order = (q desc, more recently accessed first, never accessed in symbol order)
    for (int d = lane; d < 256; d += 64) {
        const int qd = s_q[d], td = s_t[d];
        int r = 0;
        for (int e = 0; e < 256; e++) {
            const int qe = s_q[e], te = s_t[e];
            const bool above = qe > qd || (qe == qd && (te > td || (te == td && e < d)));
            r += above ? 1 : 0;
        }
        s_s2r[d] = (uint8_t)r;
        s_r2s[r] = (uint8_t)d;
    }
    wave_sync();
    if (lane == 0) {
        for (uint32_t k = 0; k < cnt; k++) {
            const int i = (int)(lo + k);
            const uint8_t c = s_in[k];
            int r = s_s2r[c];
            s_out[k] = (uint8_t)r;
            const int qc = knz_sbrt_q(a.mode, i, s_p[c]);
            s_p[c] = i;
            s_q[c] = qc;
            while (r > 0 && s_q[s_r2s[r - 1]] <= qc) {
                const uint8_t t = s_r2s[r - 1];
                s_r2s[r] = t; s_s2r[t] = (uint8_t)r;
                r--;
            }
            s_r2s[r] = c;
            s_s2r[c] = (uint8_t)r;
        }
    }
    wave_sync();
    for (uint32_t i = lane; i < cnt; i += 64) dst[lo + i] = s_out[i];
}

// SBRT inverse: one chain per block (SBRT.go:204-223), tiles staged through LDS
__global__ __launch_bounds__(64) void knz_sbrt_inverse_kernel(XfArgs a) {
    __shared__ uint8_t s_in[4096], s_out[4096];
    __shared__ uint8_t s_r2s[256];
    __shared__ int s_p[256], s_q[256];
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    if (lane == 0) { a.out_len[b] = n; a.ok[b] = n <= a.out_cap ? 1 : -KNZ_ERR_PROCESS_BLOCK; }
    if (n > a.out_cap) return;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    for (int d = lane; d < 256; d += 64) { s_r2s[d] = (uint8_t)d; s_p[d] = 0; s_q[d] = 0; }
    for (uint32_t base = 0; base < n; base += 4096) {
        const uint32_t cnt = min(4096u, n - base);
        wave_sync();
        for (uint32_t i = lane; i < cnt; i += 64) s_in[i] = src[base + i];
        wave_sync();
        if (lane == 0) {
            for (uint32_t k = 0; k < cnt; k++) {
                const int i = (int)(base + k);
                int r = s_in[k];
                const uint8_t c = s_r2s[r];
                s_out[k] = c;
                const int qc = knz_sbrt_q(a.mode, i, s_p[c]);
                s_p[c] = i;
                s_q[c] = qc;
                while (r > 0 && s_q[s_r2s[r - 1]] <= qc) { s_r2s[r] = s_r2s[r - 1]; r--; }
                s_r2s[r] = c;
            }
        }
        wave_sync();
        for (uint32_t i = lane; i < cnt; i += 64) dst[base + i] = s_out[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// ZRLT forward
// 1) per segment: block-local index of the last non-zero byte (-1 if none)
__global__ __launch_bounds__(256) void knz_zrlt_seg_lastnz_kernel(XfArgs a) {
    __shared__ int s_max;
    const int tid = threadIdx.x;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG);
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    if (tid == 0) s_max = -1;
    __syncthreads();
    int m = -1;
    for (uint32_t i = lo + tid; i < hi; i += 256) if (src[i] != 0) m = (int)i;
    if (m >= 0) atomicMax(&s_max, m);
    __syncthreads();
    if (tid == 0) a.seg_a[blockIdx.x] = s_max;
}

// element size at position i given the index of the last non-zero byte before it (ZRLT.go:76-124)
__device__ __forceinline__ uint32_t knz_zrlt_elem_size(const uint8_t* src, uint32_t n, uint32_t i, int lastNZ) {
    const uint8_t v = src[i];
    if (v != 0) return v >= 0xFE ? 2u : 1u;
    if (i + 1 < n && src[i + 1] == 0) return 0u;           // not the last zero of its run
    const uint32_t k = (uint32_t)((int)i - lastNZ);        // run length
    return 31u - (uint32_t)__builtin_clz(k + 1);           // floor(log2(k+1)) digit bytes
}

// 2) per block: exclusive max-scan of the segments' last non-zero index -> seg_a[s] = last non-zero BEFORE segment s
__global__ __launch_bounds__(64) void knz_zrlt_carry_kernel(XfArgs a) {
    const uint32_t b = blockIdx.x;
    if (!a.active[b] || threadIdx.x != 0) return;
    const uint32_t nseg = (a.in_len[b] + KNZ_SEG - 1) / KNZ_SEG;
    int carry = -1;
    for (uint32_t s = 0; s < nseg; s++) {
        int32_t* p = a.seg_a + (size_t)b * a.segs_per_block + s;
        const int v = *p;
        *p = carry;
        if (v > carry) carry = v;
    }
}

// 3) per segment: output bytes produced by the elements that END in this segment -> seg_b
// 5) (scatter = true) write them at seg_b (now the segment's output offset)
template <bool SCATTER>
__global__ __launch_bounds__(256) void knz_zrlt_seg_kernel(XfArgs a) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_carry;
    __shared__ int s_lastnz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = blockIdx.x / a.segs_per_block, s = blockIdx.x % a.segs_per_block;
    if (!a.active[b]) return;
    if (SCATTER && a.ok[b] != 1) return;
    const uint32_t n = a.in_len[b];
    const uint32_t lo = s * KNZ_SEG;
    if (lo >= n) return;
    const uint32_t hi = min(n, lo + KNZ_SEG);
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    if (tid == 0) { s_carry = SCATTER ? (uint32_t)a.seg_b[blockIdx.x] : 0u; s_lastnz = a.seg_a[blockIdx.x]; }
    __syncthreads();
    // 32 consecutive positions per thread per pass keeps the running "last non-zero" in a register
    for (uint32_t base = lo; base < hi; base += 256 * 32) {
        const uint32_t p0 = base + (uint32_t)tid * 32;
        // last non-zero before p0: scan of the per-thread maxima of the previous threads
        int myMax = -1;
        for (uint32_t j = 0; j < 32 && p0 + j < hi; j++) if (src[p0 + j] != 0) myMax = (int)(p0 + j);
        // inclusive max-scan across the workgroup
        int m = myMax;
        for (int d = 1; d < 64; d <<= 1) { int t = (int)wave_shfl((uint32_t)m, lane - d); if (lane >= d && t > m) m = t; }
        __syncthreads();
        if (lane == 63) s_wave[wave] = (uint32_t)m;
        __syncthreads();
        int before = s_lastnz;
        for (int w = 0; w < wave; w++) { int t = (int)s_wave[w]; if (t > before) before = t; }
        int exclusive = (int)wave_shfl((uint32_t)m, lane - 1);
        if (lane == 0) exclusive = -1;
        if (exclusive > before) before = exclusive;
        int passMax = (int)s_wave[0];
        for (int w = 1; w < 4; w++) if ((int)s_wave[w] > passMax) passMax = (int)s_wave[w];
        // sizes of my 32 elements
        uint32_t sz = 0;
        int lnz = before;
        for (uint32_t j = 0; j < 32 && p0 + j < hi; j++) {
            sz += knz_zrlt_elem_size(src, n, p0 + j, lnz);
            if (src[p0 + j] != 0) lnz = (int)(p0 + j);
        }
        const uint32_t incl = wave_scan_incl(sz);
        __syncthreads();
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t off = s_carry + incl - sz;
        for (int w = 0; w < wave; w++) off += s_wave[w];
        const uint32_t passTotal = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (SCATTER) {
            lnz = before;
            for (uint32_t j = 0; j < 32 && p0 + j < hi; j++) {
                const uint32_t i = p0 + j;
                const uint8_t v = src[i];
                if (v != 0) {
                    if (v >= 0xFE) { dst[off++] = 0xFF; dst[off++] = (uint8_t)(v - 0xFE); }
                    else dst[off++] = (uint8_t)(v + 1);
                    lnz = (int)i;
                } else if (!(i + 1 < n && src[i + 1] == 0)) {
                    const uint32_t run = (uint32_t)((int)i - lnz) + 1;          // runLength = k + 1 (:88)
                    uint32_t lg = 31u - (uint32_t)__builtin_clz(run);
                    while (lg > 0) { lg--; dst[off++] = (uint8_t)((run >> lg) & 1); }
                }
            }
        }
        __syncthreads();
        if (tid == 0) { s_carry += passTotal; if (passMax > s_lastnz) s_lastnz = passMax; }
        __syncthreads();
    }
    if (!SCATTER && tid == 0) a.seg_b[blockIdx.x] = (int32_t)s_carry;
}

// 4) per block: exclusive scan of the segment sizes, total, and the "would not fit in len(src)" test (ZRLT.go:93,109,118,132):
//    every write needs dstIdx < len(src) (a zero run even dstIdx + digits < len(src)), i.e. total < n, or total == n
//    when the last element is a literal.
__global__ __launch_bounds__(64) void knz_zrlt_offsets_kernel(XfArgs a) {
    const uint32_t b = blockIdx.x;
    if (!a.active[b] || threadIdx.x != 0) return;
    const uint32_t n = a.in_len[b];
    const uint32_t nseg = (n + KNZ_SEG - 1) / KNZ_SEG;
    uint64_t off = 0;
    for (uint32_t s = 0; s < nseg; s++) {
        int32_t* p = a.seg_b + (size_t)b * a.segs_per_block + s;
        const uint32_t v = (uint32_t)*p;
        *p = (int32_t)off;
        off += v;
    }
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    const bool fits = off < n || (off == n && src[n - 1] != 0);
    a.out_len[b] = (uint32_t)off;
    a.ok[b] = (fits && a.out_cap >= n) ? 1 : 0;
}

// ZRLT inverse (ZRLT.go:142-225): one chain per block. Input and output are staged through LDS tiles; lane 0 runs the
// byte automaton, all lanes refill / flush.
__global__ __launch_bounds__(64) void knz_zrlt_inverse_kernel(XfArgs a) {
    __shared__ uint8_t s_in[4096], s_out[4096];
    __shared__ uint32_t s_state[8];   // 0: inPos(consumed in tile) 1: outFill 2: pendingZeros 3: runLength acc 4: phase 5: err 6: done
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x;
    if (!a.active[b]) return;
    const uint32_t srcEnd = a.in_len[b];
    const uint32_t dstEnd = a.out_cap;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    // phases of the automaton: 0 = at element start, 1 = inside a run of digit bytes, 2 = after 0xFF escape
    uint32_t srcBase = 0, dstBase = 0;     // uniform across lanes
    if (lane == 0) { for (int i = 0; i < 8; i++) s_state[i] = 0; }
    wave_sync();
    uint32_t tileCnt = 0, inPos = 0;
    bool needRefill = true;
    for (;;) {
        if (needRefill) {
            srcBase += inPos;
            inPos = 0;
            tileCnt = min(4096u, srcEnd - srcBase);
            for (uint32_t i = lane; i < tileCnt; i += 64) s_in[i] = src[srcBase + i];
            wave_sync();
        }
        if (lane == 0) {
            uint32_t ip = inPos, of = 0, pend = s_state[2], run = s_state[3], phase = s_state[4], err = 0, done = 0;
            const uint32_t outRoom = 4096;
            for (;;) {
                if (pend > 0) {                                  // zeros still owed by the last run
                    while (pend > 0 && of < outRoom) { s_out[of++] = 0; pend--; }
                    if (pend > 0) break;                         // flush needed
                }
                if (ip >= tileCnt) {                             // input tile exhausted
                    if (srcBase + ip >= srcEnd) {                // end of input (ZRLT.go:171-173,206-222)
                        if (phase == 1) {                        // the stream ends inside a run: goto End with runLength
                            uint32_t r = run - 1;                // End: runLength-- (trailing zeros)
                            if ((uint64_t)r > (uint64_t)dstEnd - (dstBase + of)) err = 1;
                            else { pend = r; phase = 0; run = 0; if (pend > 0) continue; }
                        } else if (phase == 2) {
                            err = 0;                             // 0xFF at the very end: loop breaks with srcIdx == srcEnd
                        }
                        done = 1;
                    }
                    break;
                }
                const uint8_t v = s_in[ip];
                if (phase == 2) {                                // escaped value
                    if (dstBase + of >= dstEnd) { err = 1; done = 1; break; }
                    if (of >= outRoom) break;
                    s_out[of++] = (uint8_t)(0xFE + v);
                    ip++; phase = 0;
                    continue;
                }
                if (v <= 1) {                                    // digit byte of a run (:163-174)
                    if (phase == 0) { run = 1; phase = 1; }
                    run += run + v;
                    ip++;
                    continue;
                }
                if (phase == 1) {                                // run finished by a non-digit: emit runLength-1 zeros
                    const uint32_t r = run - 1;
                    if ((uint64_t)r >= (uint64_t)dstEnd - (dstBase + of)) { err = 1; done = 1; break; }   // :178-180
                    pend = r; phase = 0; run = 0;
                    continue;
                }
                if (dstBase + of >= dstEnd) { err = 1; done = 1; break; }
                if (of >= outRoom) break;
                if (v == 0xFF) { ip++; phase = 2; continue; }
                s_out[of++] = (uint8_t)(v - 1);
                ip++;
            }
            s_state[0] = ip; s_state[1] = of; s_state[2] = pend; s_state[3] = run; s_state[4] = phase; s_state[5] = err; s_state[6] = done;
        }
        wave_sync();
        inPos = s_state[0];
        const uint32_t of = s_state[1];
        const uint32_t done = s_state[6], err = s_state[5];
        for (uint32_t i = lane; i < of; i += 64) if (dstBase + i < dstEnd) dst[dstBase + i] = s_out[i];
        dstBase += of;
        wave_sync();
        if (done || err) {
            if (lane == 0) {
                // an escape byte left dangling or unread input is an error (:218-220)
                const bool leftover = (srcBase + inPos) < srcEnd;
                a.out_len[b] = dstBase;
                a.ok[b] = (err || leftover) ? -KNZ_ERR_PROCESS_BLOCK : 1;
            }
            return;
        }
        needRefill = inPos >= tileCnt;
    }
}

// NullTransform inside a sequence: applies, nothing moves
__global__ void knz_xf_none_kernel(uint32_t nblocks, const uint8_t* active, uint8_t* skip, uint32_t stage) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblocks && active[b]) skip[b] &= (uint8_t)~(1u << (7 - stage));
}
// inverse sequence: a stage runs for live blocks whose skip bit is clear; `live` (take) is the decode-side block liveness
__global__ void knz_xf_inv_select_kernel(uint32_t nblocks, const uint8_t* skip, const uint8_t* live, uint8_t* active, uint32_t stage, const int32_t* blk_status) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblocks) active[b] = (live[b] && blk_status[b] == 0 && !(skip[b] & (1u << (7 - stage)))) ? 1 : 0;
}
// gather of per-block byte ranges (final copy of decoded blocks / staging)
__global__ __launch_bounds__(256) void knz_copy_blocks_kernel(uint32_t nblocks, const uint64_t* src_ptr, const uint32_t* len, const uint64_t* dst_ptr, const uint8_t* live) {
    const uint32_t b = blockIdx.y;
    if (b >= nblocks || (live && !live[b])) return;
    const uint32_t n = len[b];
    const uint8_t* s = (const uint8_t*)src_ptr[b];
    uint8_t* d = (uint8_t*)dst_ptr[b];
    if (s == d) return;
    for (uint32_t i = (blockIdx.x * 256 + threadIdx.x) * 16; i < n; i += gridDim.x * 256 * 16) {
        if (i + 16 <= n && ((((uintptr_t)(s + i)) | ((uintptr_t)(d + i))) & 15) == 0) *(uint4*)(d + i) = *(const uint4*)(s + i);
        else for (uint32_t j = i; j < n && j < i + 16; j++) d[j] = s[j];
    }
}
