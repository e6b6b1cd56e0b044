// TEXT transform, data-parallel form (the one-lane scans of text.hip stay as the exact fall-back).
//
// What makes the reference's scan a chain is the dictionary: whether a word is replaced depends on which earlier words entered it. That
// dependence is CAUSAL (token t only looks at decisions of tokens < t) and sparse, so it is solved as a fixed point instead of in order:
//   guess  ins[t] = "token t enters the dictionary"                     (first guess: nobody)
//   derive P[t]   = entries made before t (a prefix sum: the entry index a word gets, the length-3 cut-off at 16384 words),
//          owner[slot] = first token that claimed the hash slot (atomicMin), the most recent entry (what slot 0 holds: every first use
//          of an entry clears slot 0, TextCodec.go:804-807)
//   re-evaluate every token in parallel against that state -> ins'[t]; repeat until ins' == ins.
// The system has exactly one fixed point (induction over t: the first token whose decision differed would have seen identical earlier
// decisions), it is the reference's sequence of events, and every round fixes at least the earliest wrong decision; real text settles in
// 3-5 rounds. Blocks that do not settle in KNZ_TCP_MAX_ROUNDS rounds, or whose dictionary would wrap (2^19 words), are left to the chain
// kernel. One workgroup of 1024 threads per block runs all phases: tokenise -> fixed point -> mark -> emit (two tile loops over the
// bytes with a running carry, the rest over tokens).

#define KNZ_TCP_THREADS 1024
#define KNZ_TCP_MAX_ROUNDS 24
#define KNZ_TCP_NIL 0x7FFFFFFF

struct TextParArgs {
    TextArgs a;
    uint32_t* tok_end;             // [nblocks * tok_stride] position of the delimiter behind the word
    uint32_t* tok_h1; uint32_t* tok_h2; uint32_t* tok_p; uint32_t* tok_ref;
    uint8_t* tok_len; uint8_t* tok_ins;
    uint64_t tok_stride;
    uint8_t* mark; uint8_t* refb;  // [nblocks * pos_stride]
    uint64_t pos_stride;
    uint32_t* ins_tok;             // [nblocks << 19] token of the r-th entry made
    unsigned long long* prof;      // [nblocks * 8] phase time stamps (KNZ_TEXT_PROF diagnostics) or null
};

#ifndef KNZ_HIP_EMU
#define KNZ_TCP_STAMP(k) do { if (pa.prof && tid == 0) pa.prof[(size_t)b * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define KNZ_TCP_STAMP(k) do { } while (0)
#endif

// exclusive prefix sum over the workgroup (16 waves); total = sum of all
__device__ __forceinline__ uint32_t knz_wg_scan_excl(uint32_t v, uint32_t* s_w, uint32_t& total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t incl = wave_scan_incl(v);
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
    for (int k = 0; k < KNZ_TCP_THREADS / 64; k++) { const uint32_t x = s_w[k]; base += k < w ? x : 0u; tot += x; }
    __syncthreads();
    total = tot;
    return base + incl - v;
}

// length of the word that ends in front of the delimiter at p (TextCodec.go:762-765), 0 = no word
__device__ __forceinline__ int knz_tcp_word_len(const uint8_t* src, int p) {
    const uint32_t c = src[p];
    if (p == 0 || knz_tc_is_text(c) || !knz_tc_is_delim(c)) return 0;
    int a = p;
    while (a > 0 && p - a <= 31 && knz_tc_is_text(src[a - 1])) a--;
    const int len = p - a;
    return (len >= 2 && len <= 31) ? len : 0;
}

struct TcpBlock {
    const uint8_t* src; const uint8_t* letters; const uint32_t* stat;
    uint32_t* tok_end; uint32_t* h1; uint32_t* h2; uint32_t* P; uint32_t* ref; uint8_t* len; uint8_t* ins;
    int32_t* owner; uint32_t* ins_tok;
    uint32_t mask; int staticSize; int static0;

    // what the slot holds when token t looks (p = entries made before t): NIL, a static entry (-1 - index) or the token that made the entry
    __device__ __forceinline__ int content(uint32_t slot, uint32_t t, uint32_t p) const {
        if (slot == 0) {
            if (p == 0) return static0;
            if (p > KNZ_TC_MAX_DICT) return KNZ_TCP_NIL;                   // (only while earlier decisions are still wrong: see the wrap check)
            const uint32_t L = ins_tok[p - 1];
            return (h1[L] & mask) == 0 ? (int)L : KNZ_TCP_NIL;
        }
        const int v = owner[slot];
        return (v < 0 || (uint32_t)v < t) ? v : KNZ_TCP_NIL;
    }
    // is entry `code` the word (hash h, length n, letters at w)? 0: pe.hash != h or another length (the reference then looks at the second
    // slot, :783-788); 1: yes; 2: hash and length agree but sameWords (from the second letter on, :791-795) fails: a hash collision, the
    // word counts as not found and the second slot is NOT consulted
    __device__ __forceinline__ int match(int code, uint32_t h, int n, const uint8_t* w, bool wsafe) const {
        if (code == KNZ_TCP_NIL) return 0;
        const uint8_t* e;
        if (code < 0) {
            const uint32_t idx = (uint32_t)(-1 - code);
            if (idx >= KNZ_TC_STATIC) return 0;                           // codec 1's escape entries: one letter
            if (stat[idx] != h || (int)(stat[KNZ_TC_STATIC + idx] >> 24) != n) return 0;
            e = letters + stat[2 * KNZ_TC_STATIC + idx];
        } else {
            if (h1[code] != h || (int)len[code] != n) return 0;
            e = src + tok_end[code] - n;                                   // (an earlier word of the block: readable wherever w is)
        }
        if (wsafe) {                                                         // 8 bytes at a time (both sides readable up to 7 bytes past the word)
            for (int k = 0; k < n; k += 8) {
                uint64_t x = knz_vle64(e + k) ^ knz_vle64(w + k);
                if (k == 0) x &= ~(uint64_t)0xFF;                                // the first letter is not compared (:793)
                if (n - k < 8) x &= ((uint64_t)1 << (8 * (n - k))) - 1;
                if (x) return 2;
            }
            return 1;
        }
        for (int k = 1; k < n; k++) if (e[k] != w[k]) return 2;
        return 1;
    }
    // P[] of the current guess, the table of who made the r-th entry and the slot owners (atomicMin), 4 tokens per thread and step;
    // returns the number of entries made
    __device__ __forceinline__ uint32_t scan_round(uint32_t nt, uint32_t* s_w) {
        uint32_t made = 0;
        for (uint32_t base = 0; base < nt; base += 4 * KNZ_TCP_THREADS) {
            const uint32_t t0 = base + 4 * threadIdx.x;
            uint32_t v = 0;
            if (t0 + 4 <= nt) v = *(const uint32_t*)(ins + t0);
            else for (uint32_t j = 0; j < 4; j++) if (t0 + j < nt) v |= (uint32_t)ins[t0 + j] << (8 * j);
            const uint32_t c = (v & 1u) + ((v >> 8) & 1u) + ((v >> 16) & 1u) + ((v >> 24) & 1u);
            uint32_t tot;
            uint32_t p = made + knz_wg_scan_excl(c, s_w, tot);
            for (uint32_t j = 0; j < 4; j++) {
                const uint32_t t = t0 + j;
                if (t >= nt) break;
                P[t] = p;
                if ((v >> (8 * j)) & 1u) {
                    if (p < KNZ_TC_MAX_DICT) ins_tok[p] = t;
                    const uint32_t slot = h1[t] & mask;
                    if (slot) atomicMin(&owner[slot], (int)t);
                    p++;
                }
            }
            made += tot;
        }
        return made;
    }
};

__global__ __launch_bounds__(KNZ_TCP_THREADS) void knz_text_forward_par_kernel(TextParArgs pa) {
    __shared__ uint32_t s_w[KNZ_TCP_THREADS / 64];
    __shared__ uint32_t s_flag;
    __shared__ uint32_t s_txt[2 + 2 * 64 + 2];                               // letter flags of a tile: 64 masks of 64 positions behind the last mask of the previous tile
    const TextArgs& a = pa.a;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    if (!a.active[b]) return;
    const int mode = a.tmode[b];
    if (mode < 0) { if (tid == 0) { a.ok[b] = mode == -2 ? 1 : 0; a.out_len[b] = 0; a.tmode[b] = -3; } return; }
    const int count = (int)a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    const uint32_t kind = a.kind;
    TcpBlock k;
    k.src = src; k.stat = a.stat; k.letters = (const uint8_t*)(a.stat + 3 * KNZ_TC_STATIC);
    k.tok_end = pa.tok_end + (size_t)b * pa.tok_stride; k.h1 = pa.tok_h1 + (size_t)b * pa.tok_stride; k.h2 = pa.tok_h2 + (size_t)b * pa.tok_stride;
    k.P = pa.tok_p + (size_t)b * pa.tok_stride; k.ref = pa.tok_ref + (size_t)b * pa.tok_stride;
    k.len = pa.tok_len + (size_t)b * pa.tok_stride; k.ins = pa.tok_ins + (size_t)b * pa.tok_stride;
    k.owner = a.dict_map + ((size_t)b << a.log_hash); k.ins_tok = pa.ins_tok + ((size_t)b << 19);
    k.mask = (1u << a.log_hash) - 1; k.staticSize = kind == 1 ? (int)KNZ_TC_STATIC + 2 : (int)KNZ_TC_STATIC;
    uint8_t* mark = pa.mark + (size_t)b * pa.pos_stride;
    uint8_t* refb = pa.refb + (size_t)b * pa.pos_stride;
    const uint32_t nslots = 1u << a.log_hash;

    // ---- phase A: tokens (words of 2..31 letters in front of a delimiter), in order, with both hashes ------------------------------
    KNZ_TCP_STAMP(0);
    // Tile of 4096 positions, thread tid looks at positions base + 1024 j + tid (j = 0..3): one ballot per wave and j is the "is a letter" mask
    // of 64 consecutive positions; the masks go to LDS (with the last one of the previous tile in front), so the length of the letter run in
    // front of a delimiter is a count of leading ones in a 32-bit window instead of a byte-by-byte walk back through memory.
    uint32_t nt = 0;
    if (tid < 2) s_txt[tid] = 0;                                            // (positions in front of the block: not letters)
    for (int base = 0; base < count; base += 4 * KNZ_TCP_THREADS) {
        uint32_t cur[4];
        for (int j = 0; j < 4; j++) {
            const int p = base + j * KNZ_TCP_THREADS + (int)tid;
            cur[j] = p < count ? src[p] : 0u;
            const uint64_t m = wave_ballot(p < count && knz_tc_is_text(cur[j]));
            if ((tid & 63) == 0) { const uint32_t wi = 2 + 2 * (uint32_t)(j * (KNZ_TCP_THREADS / 64) + (tid >> 6)); s_txt[wi] = (uint32_t)m; s_txt[wi + 1] = (uint32_t)(m >> 32); }
        }
        __syncthreads();
        int wl[4];
        uint32_t c012 = 0, c3 = 0;
        for (int j = 0; j < 4; j++) {
            const int idx = j * KNZ_TCP_THREADS + (int)tid, p = base + idx;
            wl[j] = 0;
            if (p < count && p > 0 && !knz_tc_is_text(cur[j]) && knz_tc_is_delim(cur[j])) {
                // letter flags of the 32 positions in front of p: bit 31 = position p - 1 (s_txt bit 64 = the tile's first position)
                const uint32_t lo = 64u + (uint32_t)idx - 32u, wq = lo >> 5, sh = lo & 31u;
                const uint64_t two = ((uint64_t)s_txt[wq + 1] << 32) | s_txt[wq];
                const uint32_t win = (uint32_t)(two >> sh);
                const int n = (int)__clz((int)~win);                             // leading ones (0 when p - 1 is no letter, 32 = longer than 31)
                if (n >= 2 && n <= 31) wl[j] = n;
            }
            if (j < 3) c012 += (wl[j] ? 1u : 0u) << (11 * j); else c3 = wl[j] ? 1u : 0u;
        }
        uint32_t tot012, tot3;
        const uint32_t e012 = knz_wg_scan_excl(c012, s_w, tot012), e3 = knz_wg_scan_excl(c3, s_w, tot3);
        const uint32_t tj[4] = {tot012 & 0x7FFu, (tot012 >> 11) & 0x7FFu, (tot012 >> 22) & 0x7FFu, tot3};
        uint32_t first = nt;
        for (int j = 0; j < 4; j++) {
            if (wl[j]) {
                const uint32_t t = first + (j < 3 ? (e012 >> (11 * j)) & 0x7FFu : e3);
                const int p = base + j * KNZ_TCP_THREADS + (int)tid, n = wl[j];
                const uint8_t* w = src + p - n;
                uint32_t h1 = knz_tc_hash_step(KNZ_TC_HASH1, w[0]), h2 = knz_tc_hash_step(KNZ_TC_HASH1, (uint32_t)w[0] ^ 0x20u);
                for (int q = 1; q < n; q++) { const uint32_t h = (uint32_t)w[q] * KNZ_TC_HASH2; h1 = (h1 * KNZ_TC_HASH1) ^ h; h2 = (h2 * KNZ_TC_HASH1) ^ h; }
                k.tok_end[t] = (uint32_t)p; k.len[t] = (uint8_t)n; k.h1[t] = h1; k.h2[t] = h2; k.ins[t] = 0;
            }
            first += tj[j];
        }
        nt = first;
        if (tid < 2) s_txt[tid] = s_txt[2 + 2 * 63 + tid];                   // the tile's last 64 flags lead the next tile
        __syncthreads();
    }
    // static entries claim their slots (the later entry wins a shared slot: -1 - index, smallest value = largest index)
    for (uint32_t s = tid; s < nslots; s += KNZ_TCP_THREADS) k.owner[s] = KNZ_TCP_NIL;
    for (uint32_t s = tid; s < (uint32_t)count; s += KNZ_TCP_THREADS) mark[s] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < KNZ_TC_STATIC; i += KNZ_TCP_THREADS) atomicMin(&k.owner[a.stat[i] & k.mask], -1 - (int)i);
    if (kind == 1 && tid == 0) atomicMin(&k.owner[0], -1 - ((int)KNZ_TC_STATIC + 1));
    __syncthreads();
    k.static0 = k.owner[0];
    __syncthreads();

    // first guess: the first word of at least 3 letters per free slot (what the fixed point is when nothing is found under the other-case
    // hash and the length-3 cut-off is not reached: two rounds less than starting from "nobody")
    for (uint32_t t = tid; t < nt; t += KNZ_TCP_THREADS) { const uint32_t slot = k.h1[t] & k.mask; if (k.len[t] >= 3 && slot) atomicMin(&k.owner[slot], (int)t); }
    __syncthreads();
    for (uint32_t t = tid; t < nt; t += KNZ_TCP_THREADS) { const uint32_t slot = k.h1[t] & k.mask; k.ins[t] = (k.len[t] >= 3 && slot && k.owner[slot] == (int)t) ? 1 : 0; }
    __syncthreads();
    // ---- phase B: which tokens enter the dictionary (fixed point, see the header) ---------------------------------------------------
    KNZ_TCP_STAMP(1);
    int rounds = 0;
    bool settled = false;
    uint32_t madeLast = 0;                                                   // entries made under the guess the last round started from
    for (int round = 0; round < KNZ_TCP_MAX_ROUNDS && !settled; round++) {
        rounds = round + 1;
        for (uint32_t s = tid; s < nslots; s += KNZ_TCP_THREADS) if (k.owner[s] >= 0) k.owner[s] = KNZ_TCP_NIL;
        __syncthreads();
        const uint32_t made = k.scan_round(nt, s_w);
        madeLast = made;
        if (tid == 0) s_flag = 0;
        __syncthreads();
        bool changed = false;
        for (uint32_t t = tid; t < nt; t += KNZ_TCP_THREADS) {
            const int n = k.len[t];
            const uint8_t* w = src + k.tok_end[t] - n;
            const uint32_t h1 = k.h1[t], h2 = k.h2[t], p = k.P[t];
            const int c1 = k.content(h1 & k.mask, t, p);
            int hit = KNZ_TCP_NIL;
            uint32_t via2 = 0;
            const bool wsafe = (int)k.tok_end[t] - n + 40 <= count;           // 8-byte reads stay inside the block
            const int m1 = k.match(c1, h1, n, w, wsafe);
            if (m1 == 1) hit = c1;
            else if (m1 == 0) { const int c2 = k.content(h2 & k.mask, t, p); if (k.match(c2, h2, n, w, wsafe) == 1) { hit = c2; via2 = 1; } }
            const bool qual = n > 3 || (n == 3 && k.staticSize + (int)p < 16384);
            const uint32_t nv = (hit == KNZ_TCP_NIL && qual && c1 == KNZ_TCP_NIL) ? 1u : 0u;
            if (nv != k.ins[t]) { k.ins[t] = (uint8_t)nv; changed = true; }
            // reference found: bit 31, first-letter flip: bit 30, entry code in the low bits (static index, or 2^20 + token that made it)
            k.ref[t] = hit == KNZ_TCP_NIL ? 0u : (0x80000000u | (via2 << 30) | (hit < 0 ? (uint32_t)(-1 - hit) : 0x100000u + (uint32_t)hit));
        }
        if (changed) s_flag = 1;
        __syncthreads();
        settled = s_flag == 0;
        __syncthreads();
    }
    // no fixed point within the rounds, or one in which the dictionary wraps (2^19 words: entries get recycled, TextCodec.go:816-821):
    // tmode[b] keeps the mode and the chain kernel takes the block. (Early rounds over-count freely: the first one lets every word in.)
    if (!settled || (uint32_t)k.staticSize + madeLast >= KNZ_TC_MAX_DICT) return;

    // ---- phase C: mark replaced words, their index bytes, and the lone spaces between two of them ----------------------------------
    KNZ_TCP_STAMP(2);
    if (pa.prof && tid == 0) { pa.prof[(size_t)b * 8 + 6] = (unsigned long long)rounds; pa.prof[(size_t)b * 8 + 7] = nt; }
    for (uint32_t t = tid; t < nt; t += KNZ_TCP_THREADS) {
        const uint32_t r = k.ref[t];
        if (!(r & 0x80000000u)) continue;
        const uint32_t code = r & 0x3FFFFFFFu;
        const int idx = code >= 0x100000u ? k.staticSize + (int)k.P[code - 0x100000u] : (int)code;
        const int n = k.len[t], end = (int)k.tok_end[t], start = end - n;
        uint8_t ib[3];
        const int il = kind == 1 ? knz_tc_emit_index1(ib, idx) : knz_tc_emit_index2(ib, idx);
        for (int q = 0; q < il; q++) refb[start + q] = ib[q];
        mark[start] = (uint8_t)(0x40 | ((r >> 30) & 1u) << 2 | (uint32_t)il);
        for (int q = 1; q < n; q++) mark[start + q] = 0x80;
        if (t > 0 && (k.ref[t - 1] & 0x80000000u) && (int)k.tok_end[t - 1] == start - 1 && src[start - 1] == ' ') mark[start - 1] = 0x20;
    }
    __syncthreads();

    // ---- phase D: emit (emitSymbols :884-934 / :1415-1487 + the index bytes), with the reference's room checks --------------------
    KNZ_TCP_STAMP(3);
    const bool crlf = (mode & 0x40) != 0;
    const int dstEnd = count, dstEndRef = kind == 1 ? dstEnd - 4 : dstEnd - 3;
    uint32_t outPos = 1;
    bool fail = false;
    if (tid == 0) dst[0] = (uint8_t)mode;
    for (int base = 0; base < count; base += 4 * KNZ_TCP_THREADS) {
        const int p0 = base + 4 * (int)tid;
        uint32_t cl[4], cs = 0;
        uint8_t mk[4], sb[4];
        for (int j = 0; j < 4; j++) {
            uint32_t c = 0;
            mk[j] = 0; sb[j] = 0;
            if (p0 + j < count) {
                const uint32_t m = mark[p0 + j], cur = src[p0 + j];
                mk[j] = (uint8_t)m; sb[j] = (uint8_t)cur;
                if (m & 0x40) c = (m & 3u) + (kind == 1 ? 1u : ((m >> 2) & 1u));
                else if (m & 0xA0) c = 0;
                else if (kind == 1) c = (cur == 0x0F || cur == 0x0E) ? 3u : ((cur == 0x0D && crlf) ? 0u : 1u);
                else c = cur == 0x0F ? 2u : ((cur == 0x0D && crlf) ? 0u : (cur >= 0x80 ? 2u : 1u));
            }
            cl[j] = c; cs += c;
        }
        uint32_t tot;
        uint32_t o = outPos + knz_wg_scan_excl(cs, s_w, tot);
        for (int j = 0; j < 4; j++) {
            if (p0 + j >= count) break;
            const uint32_t m = mk[j], cur = sb[j];
            if (m & 0x40) {
                if ((int)o >= dstEndRef) fail = true;
                else {
                    uint32_t q = o;
                    if (kind == 1) dst[q++] = (m & 4u) ? 0x0E : 0x0F; else if (m & 4u) dst[q++] = 0x80;
                    for (uint32_t z = 0; z < (m & 3u); z++) dst[q++] = refb[p0 + j + (int)z];
                }
            } else if (m & 0xA0) {
            } else if (kind == 1) {
                if ((int)o >= dstEnd) fail = true;                           // (checked in front of every literal, a dropped CR included)
                else if (cur == 0x0F || cur == 0x0E) {
                    if ((int)o + 3 >= dstEnd) fail = true;
                    else { const int idx = cur == 0x0F ? k.staticSize - 1 : k.staticSize - 2; dst[o] = 0x0F; dst[o + 1] = (uint8_t)(0x80 | (idx >> 7)); dst[o + 2] = (uint8_t)(idx & 0x7F); }
                } else if (cl[j]) dst[o] = (uint8_t)cur;
            } else {
                if (cur == 0x0F) { if ((int)o + 1 >= dstEnd) fail = true; else { dst[o] = 0x0F; dst[o + 1] = 0x0F; } }
                else if (cl[j] == 2) { if ((int)o + 1 >= dstEnd) fail = true; else { dst[o] = 0x0F; dst[o + 1] = (uint8_t)cur; } }
                else if (cl[j] == 1) { if ((int)o >= dstEnd) fail = true; else dst[o] = (uint8_t)cur; }
            }
            o += cl[j];
        }
        outPos += tot;
    }
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (fail) s_flag = 1;
    __syncthreads();
    KNZ_TCP_STAMP(4);
    if (tid == 0) {
        const bool bad = s_flag != 0 || (int)outPos > dstEnd;
        a.ok[b] = bad ? 0 : 1;
        a.out_len[b] = bad ? 0u : outPos;
        a.tmode[b] = -3;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Inverse, data-parallel. The encoded block is a string of items: literal bytes, escaped literals (codec 2: 0x0F + byte) and references
// (codec 1: 0x0F / 0x0E + 1..3 index bytes ; codec 2: [0x80] + a lead byte >= 0x80 + 0..2 more bytes). Index bytes can take any value, so
// where an item starts is a property of the whole prefix: a 5-state machine over the bytes.
//   phase I   the state in front of every byte: each thread composes the transition maps of its stretch of bytes (all start states at
//             once), the 1024 maps are chained, each thread replays its stretch from its true start state.
//   phase II  the literal words the decoder's dictionary sees (>= 3 letters, in front of a literal delimiter), in order, hashed.
//   phase III which of them enter the dictionary: the same causal fixed point as the encoder's, without matching (a word enters when
//             its slot is empty, TextCodec.go:1004-1027).
//   phase IV  output length of every item (a reference: the word it names, + the implied space between two word references), prefix
//             sum with a running carry, bytes written by the thread that owns the item.
// Anything a valid stream cannot contain (a reference to an entry that does not exist yet, a truncated item, output that does not fit)
// fails the block exactly where the reference's scan fails it; a dictionary that wraps goes to the chain kernel.
enum { KNZ_TS_N = 0, KNZ_TS_F = 1, KNZ_TS_I1 = 2, KNZ_TS_I2 = 3, KNZ_TS_E = 4 };     // codec 1 uses N, I* as X0 = 1, X1 = 2, X2 = 3

__device__ __forceinline__ uint32_t knz_ts_next(uint32_t kind, uint32_t s, uint32_t c) {
    if (kind == 1) {
        if (s == 0) return (c == 0x0F || c == 0x0E) ? 1u : 0u;
        if (s == 1 || s == 2) return c < 128 ? 0u : s + 1;
        return 0u;
    }
    if (s == KNZ_TS_N) {
        if (c == 0x80) return KNZ_TS_F;
        if (c > 0x80) { const uint32_t i = c & 0x7F; return i >= 112 ? KNZ_TS_I2 : (i >= 64 ? KNZ_TS_I1 : KNZ_TS_N); }
        return c == 0x0F ? KNZ_TS_E : KNZ_TS_N;
    }
    if (s == KNZ_TS_F) { const uint32_t i = c & 0x7F; return i >= 112 ? KNZ_TS_I2 : (i >= 64 ? KNZ_TS_I1 : KNZ_TS_N); }
    if (s == KNZ_TS_I2) return KNZ_TS_I1;
    return KNZ_TS_N;
}
// the byte at q (state st in front of it) is the last byte of a reference to a WORD (an entry longer than one letter)
__device__ __forceinline__ bool knz_ts_word_ref_end(uint32_t kind, const uint8_t* src, const uint8_t* st, int q) {
    if (q < 1) return false;
    const uint32_t s = st[q] & 7u, c = src[q];
    if (kind == 1) {
        if (s == 1) return c < 128;                                         // one index byte: a static word
        uint32_t idx;
        if (s == 2) { if (c >= 128) return false; idx = (((uint32_t)src[q - 1] & 0x7F) << 7) | c; }
        else if (s == 3) idx = (((((uint32_t)src[q - 2] & 0x1F) << 7) | ((uint32_t)src[q - 1] & 0x7F)) << 7) | c;
        else return false;
        return idx != KNZ_TC_STATIC && idx != KNZ_TC_STATIC + 1;             // (the two escape entries are one letter long: wordRun ends)
    }
    if (s == KNZ_TS_I1) return true;
    if (s == KNZ_TS_F) return (c & 0x7F) < 64;
    return s == KNZ_TS_N && c > 0x80 && (c & 0x7F) < 64;
}
// what the byte at p (state s in front of it) does to the decoder's wordRun flag (:1062-1075, :1093): 1 = it ends a reference to a word
// (the flag is set), 2 = it is or ends an item that clears the flag (a literal that is not a letter, an escaped literal, a reference to a
// one-letter escape entry), 0 = nothing (letters leave the flag alone; so do bytes inside an item)
__device__ __forceinline__ uint32_t knz_ts_event(uint32_t kind, uint32_t s, uint32_t c, uint32_t prev1, uint32_t prev2) {   // prev1 / prev2: the bytes at p - 1 / p - 2
    if (kind == 1) {
        if (s == 0) return (c == 0x0F || c == 0x0E || knz_tc_is_text(c)) ? 0u : 2u;
        if (s == 1) return c < 128 ? 1u : 0u;
        uint32_t idx;
        if (s == 2) { if (c >= 128) return 0u; idx = ((prev1 & 0x7F) << 7) | c; }
        else idx = ((((prev2 & 0x1F) << 7) | (prev1 & 0x7F)) << 7) | c;
        return (idx == KNZ_TC_STATIC || idx == KNZ_TC_STATIC + 1) ? 2u : 1u;
    }
    if (s == KNZ_TS_N) {
        if (c == 0x80 || c == 0x0F) return 0u;
        if (c > 0x80) return (c & 0x7F) < 64 ? 1u : 0u;
        return knz_tc_is_text(c) ? 0u : 2u;
    }
    if (s == KNZ_TS_F) return (c & 0x7F) < 64 ? 1u : 0u;
    if (s == KNZ_TS_I1) return 1u;
    if (s == KNZ_TS_E) return 2u;
    return 0u;
}
// literal word in front of the literal delimiter at e (:983-986): its length (0 = none) and where its letters start
__device__ __forceinline__ int knz_tsp_word_len(uint32_t kind, const uint8_t* src, const uint8_t* st, int e, int& start) {
    const uint32_t c = src[e];
    if ((st[e] & 7u) != KNZ_TS_N || knz_tc_is_text(c) || !knz_tc_is_delim(c)) return 0;
    int a = e;
    while (a > 1 && e - a <= 32 && (st[a - 1] & 7u) == KNZ_TS_N && knz_tc_is_text(src[a - 1])) a--;
    if (knz_ts_word_ref_end(kind, src, st, a - 1)) a++;                     // delimAnchor sits one byte behind a word reference (:1070)
    const int len = e - a;
    start = a;
    return (len >= 3 && len <= 31) ? len : 0;
}

// f(byte, position) for the bytes of [lo, hi) in order, 8 bytes per load: 1024 threads that each walk a stretch of their own touch 64 different
// lines per load instruction, so the number of load instructions is what the walk costs
template <typename F>
__device__ __forceinline__ void knz_tsp_walk(const uint8_t* src, int lo, int hi, F f) {
    int p = lo;
    for (; p + 8 <= hi; p += 8) {
        const uint64_t w = knz_vle64(src + p);
#pragma unroll
        for (int k = 0; k < 8; k++) f((uint32_t)(w >> (8 * k)) & 0xFFu, p + k);
    }
    for (; p < hi; p++) f((uint32_t)src[p], p);
}

__global__ __launch_bounds__(KNZ_TCP_THREADS) void knz_text_inverse_par_kernel(TextParArgs pa) {
    __shared__ uint32_t s_w[KNZ_TCP_THREADS / 64];
    __shared__ uint32_t s_flag;
    __shared__ uint32_t s_txt[2 + 2 * 64 + 2];                               // literal-letter flags of a tile behind the last mask of the previous tile
    __shared__ uint16_t s_map[KNZ_TCP_THREADS];
    __shared__ uint8_t s_start[KNZ_TCP_THREADS];
    const TextArgs& a = pa.a;
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    if (!a.active[b]) return;
    const int m = (int)a.in_len[b];
    const int64_t dstEnd = (int64_t)a.out_cap;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    if (m == 0 || dstEnd == 0) { if (tid == 0) { a.ok[b] = 1; a.out_len[b] = 0; a.tmode[b] = -3; } return; }
    if (m < 2 || (uint32_t)m > (1u << 30)) { if (tid == 0) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; a.out_len[b] = 0; a.tmode[b] = -3; } return; }
    const uint32_t kind = a.kind;
    TcpBlock k;
    k.src = src; k.stat = a.stat; k.letters = (const uint8_t*)(a.stat + 3 * KNZ_TC_STATIC);
    k.tok_end = pa.tok_end + (size_t)b * pa.tok_stride; k.h1 = pa.tok_h1 + (size_t)b * pa.tok_stride; k.h2 = pa.tok_h2 + (size_t)b * pa.tok_stride;
    k.P = pa.tok_p + (size_t)b * pa.tok_stride; k.ref = pa.tok_ref + (size_t)b * pa.tok_stride;
    k.len = pa.tok_len + (size_t)b * pa.tok_stride; k.ins = pa.tok_ins + (size_t)b * pa.tok_stride;
    k.owner = a.dict_map + ((size_t)b << a.log_hash); k.ins_tok = pa.ins_tok + ((size_t)b << 19);
    k.mask = (1u << a.log_hash) - 1; k.staticSize = kind == 1 ? (int)KNZ_TC_STATIC + 2 : (int)KNZ_TC_STATIC;
    uint8_t* st = pa.mark + (size_t)b * pa.pos_stride;
    const uint32_t nslots = 1u << a.log_hash;
    if (tid == 0) s_flag = 0;

    // ---- phase I: parser state in front of every byte ------------------------------------------------------------------------------
    KNZ_TCP_STAMP(0);
    const int seg = (m - 1 + KNZ_TCP_THREADS - 1) / KNZ_TCP_THREADS;
    const int lo = min(m, 1 + (int)tid * seg), hi = min(m, lo + seg);
    {
        uint32_t v0 = 0, v1 = 1, v2 = 2, v3 = 3, v4 = 4;                     // where each start state has got to
        bool merged = false;                                                 // the start state no longer matters: one walk for the rest
        knz_tsp_walk(src, lo, hi, [&](uint32_t c, int) {
            v0 = knz_ts_next(kind, v0, c);
            if (!merged) {
                v1 = knz_ts_next(kind, v1, c); v2 = knz_ts_next(kind, v2, c); v3 = knz_ts_next(kind, v3, c); v4 = knz_ts_next(kind, v4, c);
                merged = (v0 == v1) & (v1 == v2) & (v2 == v3) & (v3 == v4);
            }
        });
        if (merged) v1 = v2 = v3 = v4 = v0;
        s_map[tid] = (uint16_t)(v0 | (v1 << 3) | (v2 << 6) | (v3 << 9) | (v4 << 12));
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t s = KNZ_TS_N;
        for (int i = 0; i < KNZ_TCP_THREADS; i++) { s_start[i] = (uint8_t)s; s = ((uint32_t)s_map[i] >> (3 * s)) & 7u; }
        if (s != KNZ_TS_N) s_flag = 1;                                       // the block ends inside an item
    }
    __syncthreads();
    const uint32_t pv1 = lo >= 1 ? src[lo - 1] : 0u, pv2 = lo >= 2 ? src[lo - 2] : 0u;   // the two bytes in front of the stretch (index bytes of a reference that straddles it)
    {   // wordRun in front of every byte: the last setter / clearer wins, letters pass it on (so it can reach across any number of them)
        uint32_t s = s_start[tid], ev = 0, p1 = pv1, p2 = pv2;
        knz_tsp_walk(src, lo, hi, [&](uint32_t c, int) {
            const uint32_t e = knz_ts_event(kind, s, c, p1, p2);
            ev = e ? e : ev;
            s = knz_ts_next(kind, s, c);
            p2 = p1; p1 = c;
        });
        s_map[tid] = (uint16_t)ev;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t wr = 0;
        for (int i = 0; i < KNZ_TCP_THREADS; i++) { const uint32_t ev = s_map[i]; s_start[i] = (uint8_t)(s_start[i] | (wr << 3)); wr = ev ? (ev == 1 ? 1u : 0u) : wr; }
    }
    __syncthreads();
    {
        uint32_t s = s_start[tid] & 7u, wr = s_start[tid] >> 3, p1 = pv1, p2 = pv2;
        uint64_t acc = 0;                                                    // 8 state bytes per store
        knz_tsp_walk(src, lo, hi, [&](uint32_t c, int p) {
            const uint32_t v = s | (wr << 3);
            const int k = (p - lo) & 7;
            acc |= (uint64_t)v << (8 * k);
            if (k == 7) { ((KnzPacked64*)(st + p - 7))->v = acc; acc = 0; }
            const uint32_t e = knz_ts_event(kind, s, c, p1, p2);
            wr = e ? (e == 1 ? 1u : 0u) : wr;
            s = knz_ts_next(kind, s, c);
            p2 = p1; p1 = c;
        });
        const int rest = (hi - lo) & 7;                                      // the last < 8 states of the stretch
        for (int k = 0; k < rest; k++) st[hi - rest + k] = (uint8_t)(acc >> (8 * k));
        if (tid == 0) st[0] = 7;
    }
    __syncthreads();
    if (s_flag) { if (tid == 0) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; a.out_len[b] = 0; a.tmode[b] = -3; } return; }

    // ---- phase II: literal words ------------------------------------------------------------------------------------------------------
    KNZ_TCP_STAMP(1);
    // As in the forward kernel: tile of 4096 positions, thread tid looks at positions base + 1024 j + tid, one ballot per wave and j = the
    // "literal letter" flags (a letter read in the normal state) of 64 consecutive positions, kept in LDS behind the last mask of the tile before.
    uint32_t nt = 0;
    if (tid < 2) s_txt[tid] = 0;
    for (int base = 0; base < m; base += 4 * KNZ_TCP_THREADS) {
        uint32_t cur[4], stv[4];
        for (int j = 0; j < 4; j++) {
            const int p = base + j * KNZ_TCP_THREADS + (int)tid;
            cur[j] = p < m ? src[p] : 0u;
            stv[j] = p < m ? st[p] : 7u;
            const uint64_t mk = wave_ballot(p < m && (stv[j] & 7u) == KNZ_TS_N && knz_tc_is_text(cur[j]));
            if ((tid & 63) == 0) { const uint32_t wi = 2 + 2 * (uint32_t)(j * (KNZ_TCP_THREADS / 64) + (tid >> 6)); s_txt[wi] = (uint32_t)mk; s_txt[wi + 1] = (uint32_t)(mk >> 32); }
        }
        __syncthreads();
        int wl[4], ws[4];
        uint32_t c012 = 0, c3 = 0;
        for (int j = 0; j < 4; j++) {
            const int idx = j * KNZ_TCP_THREADS + (int)tid, e = base + idx;
            wl[j] = 0; ws[j] = 0;
            if (e < m && e >= 1 && (stv[j] & 7u) == KNZ_TS_N && !knz_tc_is_text(cur[j]) && knz_tc_is_delim(cur[j])) {
                // flags of the 33 positions in front of e (bit 32 = position e - 1): a run of 33 is too long whatever comes in front of it
                const uint32_t lo33 = 64u + (uint32_t)idx - 33u, wq = lo33 >> 5, sh = lo33 & 31u;
                const uint64_t two = ((uint64_t)s_txt[wq + 1] << 32) | s_txt[wq];
                uint64_t win = two >> sh;
                if (sh) win |= (uint64_t)s_txt[wq + 2] << (64 - sh);
                win &= 0x1FFFFFFFFull;
                const int run = (int)__clzll((long long)~(win << 31));             // leading ones from bit 32 down
                if (run >= 3 && run <= 32) {
                    int a0 = e - run;
                    if (knz_ts_word_ref_end(kind, src, st, a0 - 1)) a0++;         // delimAnchor sits one byte behind a word reference (:1070)
                    const int n = e - a0;
                    if (n >= 3 && n <= 31) { wl[j] = n; ws[j] = a0; }
                }
            }
            if (j < 3) c012 += (wl[j] ? 1u : 0u) << (11 * j); else c3 = wl[j] ? 1u : 0u;
        }
        uint32_t tot012, tot3;
        const uint32_t e012 = knz_wg_scan_excl(c012, s_w, tot012), e3 = knz_wg_scan_excl(c3, s_w, tot3);
        const uint32_t tj[4] = {tot012 & 0x7FFu, (tot012 >> 11) & 0x7FFu, (tot012 >> 22) & 0x7FFu, tot3};
        uint32_t first = nt;
        for (int j = 0; j < 4; j++) {
            if (wl[j]) {
                const uint32_t t = first + (j < 3 ? (e012 >> (11 * j)) & 0x7FFu : e3);
                const int n = wl[j];
                const uint8_t* w = src + ws[j];
                uint32_t h1 = KNZ_TC_HASH1;
                for (int q = 0; q < n; q++) h1 = knz_tc_hash_step(h1, w[q]);
                k.tok_end[t] = (uint32_t)(ws[j] + n); k.len[t] = (uint8_t)n; k.h1[t] = h1; k.ins[t] = 0;
                k.h2[t] = (uint32_t)(base + j * KNZ_TCP_THREADS + (int)tid);          // h2: the delimiter's position
            }
            first += tj[j];
        }
        nt = first;
        if (tid < 2) s_txt[tid] = s_txt[2 + 2 * 63 + tid];
        __syncthreads();
    }
    for (uint32_t s = tid; s < nslots; s += KNZ_TCP_THREADS) k.owner[s] = KNZ_TCP_NIL;
    __syncthreads();
    for (uint32_t i = tid; i < KNZ_TC_STATIC; i += KNZ_TCP_THREADS) atomicMin(&k.owner[a.stat[i] & k.mask], -1 - (int)i);
    if (kind == 1 && tid == 0) atomicMin(&k.owner[0], -1 - ((int)KNZ_TC_STATIC + 1));
    __syncthreads();
    k.static0 = k.owner[0];
    __syncthreads();

    for (uint32_t t = tid; t < nt; t += KNZ_TCP_THREADS) { const uint32_t slot = k.h1[t] & k.mask; if (slot) atomicMin(&k.owner[slot], (int)t); }   // first guess: the first word per free slot
    __syncthreads();
    for (uint32_t t = tid; t < nt; t += KNZ_TCP_THREADS) { const uint32_t slot = k.h1[t] & k.mask; k.ins[t] = (slot && k.owner[slot] == (int)t) ? 1 : 0; }
    __syncthreads();
    // ---- phase III: entries ------------------------------------------------------------------------------------------------------------
    KNZ_TCP_STAMP(2);
    bool settled = false;
    uint32_t madeLast = 0;
    for (int round = 0; round < KNZ_TCP_MAX_ROUNDS && !settled; round++) {
        for (uint32_t s = tid; s < nslots; s += KNZ_TCP_THREADS) if (k.owner[s] >= 0) k.owner[s] = KNZ_TCP_NIL;
        __syncthreads();
        const uint32_t made = k.scan_round(nt, s_w);
        madeLast = made;
        if (tid == 0) s_flag = 0;
        __syncthreads();
        bool changed = false;
        for (uint32_t t = tid; t < nt; t += KNZ_TCP_THREADS) {
            const int n = k.len[t];
            const uint32_t p = k.P[t];
            const int c1 = k.content(k.h1[t] & k.mask, t, p);
            const bool qual = n > 3 || k.staticSize + (int)p < 16384;
            const uint32_t nv = (qual && c1 == KNZ_TCP_NIL) ? 1u : 0u;
            if (nv != k.ins[t]) { k.ins[t] = (uint8_t)nv; changed = true; }
        }
        if (changed) s_flag = 1;
        __syncthreads();
        settled = s_flag == 0;
        __syncthreads();
    }
    if (!settled || (uint32_t)k.staticSize + madeLast >= KNZ_TC_MAX_DICT) return;       // the chain kernel takes the block

    // ---- phase IV: items -> bytes ------------------------------------------------------------------------------------------------------
    KNZ_TCP_STAMP(3);
    if (pa.prof && tid == 0) pa.prof[(size_t)b * 8 + 7] = nt;
    const bool crlf = (src[0] & 0x40) != 0;
    uint64_t outPos = 0;
    bool bad = false;
    for (int base = 0; base < m; base += 4 * KNZ_TCP_THREADS) {
        const int p0 = base + 4 * (int)tid;
        uint32_t il[4], cs = 0;                                                // output bytes of the item that starts at p0 + j
        uint32_t iw[4];                                                        // reference: entry code | flip << 30 | space << 29 | 1 << 31
        for (int j = 0; j < 4; j++) {
            const int p = p0 + j;
            il[j] = 0; iw[j] = 0;
            if (p < 1 || p >= m) continue;
            const uint32_t s = st[p] & 7u, c = src[p];
            int lead = -1;                                                     // position of the first index byte of a reference anchored here
            uint32_t flip = 0;
            int first = p;                                                     // first byte of the item
            if (kind == 1) {
                if (s == 0) { if (c == 0x0F || c == 0x0E) { lead = p + 1; flip = c == 0x0E ? 1u : 0u; } else il[j] = (crlf && c == 0x0A) ? 2u : 1u; }
            } else {
                if (s == KNZ_TS_N) { if (c > 0x80) lead = p; else if (c != 0x80 && c != 0x0F) il[j] = (crlf && c == 0x0A) ? 2u : 1u; }
                else if (s == KNZ_TS_F) { lead = p; flip = 1; first = p - 1; }
                else if (s == KNZ_TS_E) il[j] = 1;
            }
            if (lead < 0) continue;
            int idx = -1;
            if (kind == 1) {
                if (lead < m) {
                    idx = src[lead];
                    if (idx >= 128) {
                        idx &= 0x7F;
                        int idx2 = lead + 1 < m ? (int)src[lead + 1] : 0;
                        if (idx2 >= 0x80) { idx = ((idx & 0x1F) << 7) | (idx2 & 0x7F); idx2 = lead + 2 < m ? (int)src[lead + 2] : 0; }
                        idx = (idx << 7) | idx2;
                    }
                }
            } else {
                idx = (int)(c & 0x7F);
                if (idx >= 112) idx = ((idx & 0x0F) << 16) | ((lead + 1 < m ? (int)src[lead + 1] : 0) << 8) | (lead + 2 < m ? (int)src[lead + 2] : 0);
                else if (idx >= 64) idx = ((idx & 0x1F) << 8) | (lead + 1 < m ? (int)src[lead + 1] : 0);
                idx--;                                                         // (0 -> -1: invalid)
            }
            // the entry must exist when the scan gets here: static, or made by a word whose delimiter lies in front of this item
            uint32_t code = 0xFFFFFFFFu;
            int n = 0;
            if (idx >= 0 && idx < k.staticSize) { code = (uint32_t)idx; n = idx < (int)KNZ_TC_STATIC ? (int)(a.stat[KNZ_TC_STATIC + idx] >> 24) : 1; }
            else if (idx >= k.staticSize && (uint32_t)(idx - k.staticSize) < madeLast) {
                const uint32_t tk = k.ins_tok[idx - k.staticSize];
                if ((int)k.h2[tk] < first) { code = 0x100000u + tk; n = k.len[tk]; }
            }
            if (code == 0xFFFFFFFFu) { bad = true; continue; }
            const uint32_t sp = (n > 1 && (st[first] >> 3) != 0) ? 1u : 0u;    // wordRun when the item starts
            il[j] = (uint32_t)n + sp;
            iw[j] = 0x80000000u | (flip << 30) | (sp << 29) | code;
        }
        for (int j = 0; j < 4; j++) cs += il[j];
        uint32_t tot;
        uint64_t o = outPos + knz_wg_scan_excl(cs, s_w, tot);
        for (int j = 0; j < 4; j++) {
            const int p = p0 + j;
            if (p < 1 || p >= m) continue;
            const uint32_t s = st[p] & 7u, c = src[p];
            if (s == KNZ_TS_N && (int64_t)o >= dstEnd) { bad = true; o += il[j]; continue; }   // a loop iteration starts here: the scan stops at a full buffer with input left
            if (il[j] == 0) continue;
            if (iw[j]) {
                const uint32_t w = iw[j], code = w & 0x1FFFFFFFu;
                const int n = (int)il[j] - (int)((w >> 29) & 1u);
                uint64_t q = o;
                if ((w >> 29) & 1u) dst[q++] = ' ';
                if ((int64_t)(q + (uint64_t)n) >= dstEnd) { bad = true; }
                else {
                    if (code >= 0x100000u) { const uint32_t tk = code - 0x100000u; const uint8_t* e = src + k.tok_end[tk] - n; for (int z = 0; z < n; z++) dst[q + z] = e[z]; }
                    else if (code < KNZ_TC_STATIC) { const uint8_t* e = k.letters + a.stat[2 * KNZ_TC_STATIC + code]; for (int z = 0; z < n; z++) dst[q + z] = e[z]; }
                    else dst[q] = code == KNZ_TC_STATIC ? 0x0E : 0x0F;
                    if ((w >> 30) & 1u) dst[q] ^= 0x20;
                }
            } else if (il[j] == 2) {
                dst[o] = 0x0D;
                if ((int64_t)o + 1 >= dstEnd) bad = true; else dst[o + 1] = 0x0A;
            } else if (il[j] == 1) dst[o] = (uint8_t)c;
            o += il[j];
        }
        outPos += tot;
    }
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (bad) s_flag = 1;
    __syncthreads();
    KNZ_TCP_STAMP(4);
    if (tid == 0) {
        const bool err = s_flag != 0 || (int64_t)outPos > dstEnd;
        a.ok[b] = err ? -KNZ_ERR_PROCESS_BLOCK : 1;
        a.out_len[b] = err ? 0u : (uint32_t)outPos;
        a.tmode[b] = -3;
    }
}
