// TEXT transform (kanzi-go v2/transform/TextCodec.go): words found in a dictionary (1024 static English words + the words met so far
// in the block) are replaced by an index. Two stream formats behind one transform id: "codec 1" (escape bytes 0x0F / 0x0E + a 1..3
// byte index, :692-1116) and "codec 2" (indexes as bytes >= 0x80, :1225-1718), chosen from the entropy stage (Factory.go:100-120).
//
// The dictionary is a hash table slot -> entry that both directions update while they scan: a chain through the whole block. The
// kernels below keep the reference's order of events exactly:
//   * knz_text_stats_kernel: the mode byte of every block (computeTextStats :187-306, detectTextType :308-397, DetectSimpleType
//     internal/Global.go:346-420) with one workgroup per block; sets ctx["dataType"] like the reference does.
//   * knz_text_forward_chain_kernel / knz_text_inverse_chain_kernel: the scan itself, one lane per block (blocks run side by side).
// Words longer than 31 letters, the re-use of slot 0 by fresh entries, the length-3 cut-off at 16384 words and the wrap of the
// dictionary at 2^19 words are all events of that scan and come out as in the reference.

#define KNZ_TC_MAX_DICT (1u << 19)
#define KNZ_TC_STATIC 1024u
#define KNZ_TC_LETTERS_PAD 5632u               // the 5487 letters of the static dictionary, padded
#define KNZ_TC_HASH1 0x7FEB352Du
#define KNZ_TC_HASH2 0x846CA68Bu
#define KNZ_TC_PTR_STATIC 0x80000000u          // entry text lives in the static letters (else: offset into the block being scanned)
#define KNZ_TC_PTR_ESCAPE 0xC0000000u          // codec 1's two one-byte entries: the byte itself in the low bits

enum { KNZ_DT_UNDEFINED = 0, KNZ_DT_TEXT = 1, KNZ_DT_MULTIMEDIA = 2, KNZ_DT_EXE = 3, KNZ_DT_NUMERIC = 4, KNZ_DT_BASE64 = 5, KNZ_DT_DNA = 6,
       KNZ_DT_BIN = 7, KNZ_DT_UTF8 = 8, KNZ_DT_SMALL_ALPHABET = 9 };     // internal/Global.go:26-40

struct TextArgs {
    uint32_t nblocks;
    const uint64_t* in_ptr; const uint32_t* in_len;
    const uint64_t* out_ptr; uint32_t out_cap;
    uint32_t* out_len; int32_t* ok; const uint8_t* active;
    uint8_t* blk_dt;               // [nblocks] ctx["dataType"] (reference numbering); may be null (= undefined, not recorded)
    int32_t* tmode;                // [nblocks] forward: the block's mode byte, -1 = the stage declines
    int32_t* dict_map;             // [nblocks << log_hash] slot -> entry index, 0xFF-filled by the host
    uint32_t* ent;                 // [nblocks][3][2^19]: hash | length << 24 + index | text pointer
    const uint32_t* stat;          // static dictionary: hash[1024] | data[1024] | offset[1024] | letters (lower case)
    uint32_t log_hash, kind;       // kind 1 / 2 = the stream format
    uint32_t* chain_count;         // blocks scanned by the one-lane kernels (diagnostic counter)
};

__device__ __forceinline__ bool knz_tc_is_text(uint32_t v) { v |= 0x20; return v >= 'a' && v <= 'z'; }     // :492-494
__device__ __forceinline__ bool knz_tc_is_delim(uint32_t v) {                                               // :409-448
    if (v >= ' ' && v <= '/') return true;
    if (v >= ':' && v <= '?') return true;
    return v == '\n' || v == '\r' || v == '\t' || v == '_' || v == '|' || v == '{' || v == '}' || v == '[' || v == ']';
}
__device__ __forceinline__ uint32_t knz_tc_hash_step(uint32_t h, uint32_t c) { return (h * KNZ_TC_HASH1) ^ (c * KNZ_TC_HASH2); }
__device__ __forceinline__ uint32_t knz_tc_log2(uint32_t x) { return 31u - (uint32_t)__clz((int)x); }

// internal/Magic.go:83-126 GetMagicType (0 = NO_MAGIC) and the three classes the stream writer maps to a data type (:130-222)
__device__ __forceinline__ uint32_t knz_magic_type(const uint8_t* s, uint32_t n) {
    if (n < 4) return 0;
    const uint32_t key = ((uint32_t)s[0] << 24) | ((uint32_t)s[1] << 16) | ((uint32_t)s[2] << 8) | s[3];
    if ((key & ~0x0Fu) == 0xFFD8FFE0u) return key;
    if ((key >> 8) == 0x425A68u || (key >> 8) == 0x494433u) return key >> 8;
    switch (key) {
        case 0x47494638u: case 0x25504446u: case 0x504B0304u: case 0x377ABCAFu: case 0x89504E47u: case 0x7F454C46u: case 0xFEEDFACEu: case 0xCEFAEDFEu:
        case 0xFEEDFACFu: case 0xCFFAEDFEu: case 0x28B52FFDu: case 0x81CFB2CEu: case 0x4D534346u: case 0x52494646u: case 0x664C6143u: case 0xFD377A58u:
        case 0x4B414E5Au: case 0x52617221u: return key;
        default: break;
    }
    const uint32_t k16 = key >> 16, sub = (key >> 8) & 0xFF;
    if (k16 == 0x1F8Bu || k16 == 0x424Du || k16 == 0x4D5Au) return k16;
    if ((k16 == 0x5034u || k16 == 0x5035u || k16 == 0x5036u) && (sub == 0x07 || sub == 0x0A || sub == 0x0D || sub == 0x20)) return k16;
    return 0;
}
__device__ __forceinline__ uint32_t knz_magic_data_type(uint32_t m) {          // v2/io/CompressedStream.go:811-819
    switch (m) {                                                                  // IsDataCompressed
        case 0xFFD8FFE0u: case 0x47494638u: case 0x89504E47u: case 0x377ABCAFu: case 0x28B52FFDu: case 0x81CFB2CEu: case 0x4D534346u: case 0x504B0304u:
        case 0x1F8Bu: case 0x425A68u: case 0x664C6143u: case 0x494433u: case 0xFD377A58u: case 0x4B414E5Au: case 0x52617221u: return KNZ_DT_BIN;
        default: break;
    }
    switch (m) {                                                                  // IsDataMultimedia (those not already counted as compressed)
        case 0x52494646u: case 0x424Du: case 0x5034u: case 0x5035u: case 0x5036u: return KNZ_DT_MULTIMEDIA;
        default: break;
    }
    switch (m) {                                                                  // IsDataExecutable
        case 0x7F454C46u: case 0x4D5Au: case 0xFEEDFACEu: case 0xCEFAEDFEu: case 0xFEEDFACFu: case 0xCFFAEDFEu: return KNZ_DT_EXE;
        default: return KNZ_DT_UNDEFINED;
    }
}

// ctx["dataType"] of every block from its magic number, before the first transform (CompressedStream.go:811-819)
__global__ void knz_block_datatype_kernel(uint32_t nblocks, const uint64_t* blk_off, const uint32_t* blk_len, uint8_t* blk_dt) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    blk_dt[b] = (uint8_t)knz_magic_data_type(knz_magic_type((const uint8_t*)blk_off[b], blk_len[b]));
}

// ---- text statistics ------------------------------------------------------------------------------------------------------
// The byte histogram and the rows / column of the pair histogram that the decision reads: row '&', row CR, column LF and the rows
// of the UTF-8 lead bytes C2..F4. Row index of a previous byte in that table, -1 = not kept.
__device__ __forceinline__ int knz_tc_row(uint32_t prv) {
    if (prv >= 0xC2 && prv <= 0xF4) return (int)prv - 0xC2;           // 0..50
    if (prv == '&') return 51;
    if (prv == 0x0D) return 52;
    return -1;
}
#define KNZ_TC_ROWS 53

__global__ __launch_bounds__(256) void knz_text_stats_kernel(TextArgs a) {
    __shared__ uint32_t s_f0[8][256];
    __shared__ uint32_t s_rows[KNZ_TC_ROWS][256];
    __shared__ uint32_t s_colLF[256];
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    if (!a.active[b]) return;
    const uint32_t count = a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    const uint32_t dt0 = a.blk_dt ? a.blk_dt[b] : KNZ_DT_UNDEFINED;
    // TextCodec.Forward :550-568 and the delegate's first checks (:695-708): an empty block "applies", small / huge blocks and most known
    // data types decline without touching ctx["dataType"]
    if (count == 0) { if (tid == 0) a.tmode[b] = -2; return; }
    if (count < 1024u || count > (1u << 30) || a.out_cap < count || (dt0 != KNZ_DT_UNDEFINED && dt0 != KNZ_DT_TEXT && dt0 != KNZ_DT_BIN)) {
        if (tid == 0) a.tmode[b] = -1;
        return;
    }
    const bool strict = a.kind == 1;
    if (!strict && knz_magic_type(src, count) != 0) {                    // :188-192: mode = NOT_TEXT, data type bits 0
        if (tid == 0) { a.tmode[b] = -1; if (a.blk_dt) a.blk_dt[b] = KNZ_DT_UNDEFINED; }
        return;
    }
    for (uint32_t i = tid; i < 8 * 256; i += 256) (&s_f0[0][0])[i] = 0;
    for (uint32_t i = tid; i < KNZ_TC_ROWS * 256; i += 256) (&s_rows[0][0])[i] = 0;
    s_colLF[tid] = 0;
    __syncthreads();
    for (uint32_t i = tid; i < count; i += 256) {
        const uint32_t cur = src[i], prv = i ? src[i - 1] : 0u;
        atomicAdd(&s_f0[tid & 7][cur], 1u);
        const int r = knz_tc_row(prv);
        if (r >= 0) atomicAdd(&s_rows[r][cur], 1u);
        if (cur == 0x0A) atomicAdd(&s_colLF[prv], 1u);
    }
    __syncthreads();
    { uint32_t s = 0; for (int k = 0; k < 8; k++) s += s_f0[k][tid]; __syncthreads(); s_f0[0][tid] = s; }
    __syncthreads();
    if (tid != 0) return;
    const uint32_t* f0 = s_f0[0];
    const int n = (int)count;
    int nbText = (int)f0[0x0D] + (int)f0[0x0A], nbASCII = 0;
    for (int i = 0; i < 128; i++) { if (knz_tc_is_text((uint32_t)i)) nbText += (int)f0[i]; nbASCII += (int)f0[i]; }
    const int nbBin = n - nbASCII;
    bool notText;
    if (nbBin > (n >> 2)) notText = true;
    else {
        notText = nbText < n / 4;
        if (strict) notText = notText || (int)f0[0] >= n / 100 || (nbASCII / 95) < (n / 100);
        else notText = notText || (int)f0[32] < n / 50;
    }
    if (notText) {                                                       // detectTextType
        int dt = KNZ_DT_UNDEFINED;
        {   // DetectSimpleType
            int sum = 0;
            const char* dna = "acgntuACGNTU";
            for (int i = 0; i < 12; i++) sum += (int)f0[(uint8_t)dna[i]];
            if (sum > n - n / 12) dt = KNZ_DT_DNA;
            if (dt == KNZ_DT_UNDEFINED) {
                const char* num = "0123456789+-*/=,.:; ";
                sum = 0;
                for (int i = 0; i < 20; i++) sum += (int)f0[(uint8_t)num[i]];
                if (sum == n) dt = KNZ_DT_NUMERIC;
            }
            if (dt == KNZ_DT_UNDEFINED) {
                sum = (int)f0['+'] + (int)f0['/'];
                for (int c = 'A'; c <= 'Z'; c++) sum += (int)f0[c] + (int)f0[c + 32];
                for (int c = '0'; c <= '9'; c++) sum += (int)f0[c];
                if (sum + (int)f0[0x3D] == n) dt = KNZ_DT_BASE64;
            }
            if (dt == KNZ_DT_UNDEFINED) {
                sum = 0;
                for (int i = 0; i < 256; i++) sum += f0[i] > 0 ? 1 : 0;
                if (sum == 256) dt = KNZ_DT_BIN; else if (sum <= 4) dt = KNZ_DT_SMALL_ALPHABET;
            }
        }
        if (dt == KNZ_DT_UNDEFINED) {                                    // every pair a legal UTF-8 start, enough continuation bytes => UTF8
            int sum = (int)f0[0xC0] + (int)f0[0xC1];
            for (int i = 0xF5; i < 256; i++) sum += (int)f0[i];
            int sum2 = 0;
            if (sum == 0) {
                for (int i = 0; i < 256 && sum == 0; i++) {
                    if (i < 0xA0 || i > 0xBF) sum += (int)s_rows[0xE0 - 0xC2][i];
                    if (i < 0x80 || i > 0x9F) sum += (int)s_rows[0xED - 0xC2][i];
                    if (i < 0x90 || i > 0xBF) sum += (int)s_rows[0xF0 - 0xC2][i];
                    if (i < 0x80 || i > 0x8F) sum += (int)s_rows[0xF4 - 0xC2][i];
                    if (i < 0x80 || i > 0xBF) {
                        for (int j = 0xC2; j <= 0xDF; j++) sum += (int)s_rows[j - 0xC2][i];
                        for (int j = 0xE1; j <= 0xEC; j++) sum += (int)s_rows[j - 0xC2][i];
                        sum += (int)s_rows[0xF1 - 0xC2][i] + (int)s_rows[0xF2 - 0xC2][i] + (int)s_rows[0xF3 - 0xC2][i] + (int)s_rows[0xEE - 0xC2][i] +
                               (int)s_rows[0xEF - 0xC2][i];
                    } else sum2 += (int)f0[i];
                }
                if (sum == 0 && sum2 >= n / 8) dt = KNZ_DT_UTF8;
            }
        }
        a.tmode[b] = -1;
        if (a.blk_dt) a.blk_dt[b] = (uint8_t)dt;
        return;
    }
    uint32_t res = 0;
    if (nbBin <= n - n / 10) {                                           // XML / HTML flag (:256-283)
        const int f1 = (int)f0['<'], f2 = (int)f0['>'];
        const int f3 = (int)s_rows[51]['a'] + (int)s_rows[51]['g'] + (int)s_rows[51]['l'] + (int)s_rows[51]['q'];
        int minFreq = (n - nbBin) >> 9;
        if (minFreq < 2) minFreq = 2;
        if (f1 >= minFreq && f2 >= minFreq && f3 > 0) {
            if (f1 < f2) { if (f1 >= f2 - f2 / 100) res |= 0x20; }
            else if (f2 < f1) { if (f2 >= f1 - f1 / 100) res |= 0x20; }
            else res |= 0x20;
        }
    }
    if (f0[0x0D] != 0 && f0[0x0D] == f0[0x0A]) {                         // every CR is followed by LF and every LF preceded by CR (:285-303)
        bool crlf = true;
        for (int i = 0; i < 256 && crlf; i++) {
            if (i != 0x0A && s_rows[52][i] != 0) crlf = false;
            if (i != 0x0D && s_colLF[i] != 0) crlf = false;
        }
        if (crlf) res |= 0x40;
    }
    a.tmode[b] = (int32_t)res;
    if (a.blk_dt) a.blk_dt[b] = KNZ_DT_TEXT;
}

// ---- the dictionary of one block ------------------------------------------------------------------------------------------
struct TextDict {
    int32_t* map; uint32_t* eh; uint32_t* ed; uint32_t* ep;
    const uint8_t* letters; const uint8_t* text;                         // static letters ; the buffer dynamic entries point into
    uint32_t mask; int dictSize, staticSize, words; bool wrapped;

    __device__ __forceinline__ uint32_t ent_byte(uint32_t ptr, int k) const {
        if (ptr >= KNZ_TC_PTR_ESCAPE) return ptr & 0xFF;
        if (ptr & KNZ_TC_PTR_STATIC) return letters[(ptr & 0x7FFFFFFFu) + k];
        return text[ptr + k];
    }
    // sameWords(pe.ptr[1:length], word[1:]) (:399-407)
    __device__ __forceinline__ bool same_tail(uint32_t ptr, const uint8_t* w, int length) const {
        for (int k = 1; k < length; k++) if (ent_byte(ptr, k) != w[k]) return false;
        return true;
    }
    // :801-822 / :1007-1027: the next entry takes the word; a recycled entry leaves its old slot, a fresh one (hash 0) clears slot 0
    __device__ __forceinline__ void insert(uint32_t h1, int length, uint32_t textOff) {
        const uint32_t oldHash = wrapped ? eh[words] : 0u;
        map[oldHash & mask] = -1;
        eh[words] = h1; ed[words] = ((uint32_t)length << 24) | (uint32_t)words; ep[words] = textOff;
        map[h1 & mask] = words;
        words++;
        if (words >= dictSize) {
            if (dictSize >= (int)KNZ_TC_MAX_DICT) { words = staticSize; wrapped = true; }
            else dictSize <<= 1;
        }
    }
};

// reset (:652-690 / :1190-1223) by the whole wave: the static entries (and codec 1's two escape entries) enter the table, later
// entries win their slot
__device__ __forceinline__ void knz_tc_dict_reset(TextDict& d, const TextArgs& a, uint32_t b, int count, const uint8_t* text, int lane) {
    d.map = a.dict_map + ((size_t)b << a.log_hash);
    d.eh = a.ent + (size_t)b * 3 * KNZ_TC_MAX_DICT; d.ed = d.eh + KNZ_TC_MAX_DICT; d.ep = d.ed + KNZ_TC_MAX_DICT;
    d.letters = (const uint8_t*)(a.stat + 3 * KNZ_TC_STATIC); d.text = text;
    d.mask = (1u << a.log_hash) - 1;
    d.dictSize = 1 << 13;
    if (count >= 1024) { uint32_t lg = knz_tc_log2((uint32_t)count / 128); lg = lg > 18 ? 18 : (lg < 13 ? 13 : lg); d.dictSize = 1 << lg; }
    d.staticSize = a.kind == 1 ? (int)KNZ_TC_STATIC + 2 : (int)KNZ_TC_STATIC;
    d.words = d.staticSize; d.wrapped = false;
    for (uint32_t i = (uint32_t)lane; i < KNZ_TC_STATIC; i += 64) {
        const uint32_t h = a.stat[i];
        d.eh[i] = h; d.ed[i] = a.stat[KNZ_TC_STATIC + i]; d.ep[i] = KNZ_TC_PTR_STATIC | a.stat[2 * KNZ_TC_STATIC + i];
        atomicMax(&d.map[h & d.mask], (int32_t)i);
    }
    if (a.kind == 1 && lane < 2) {
        const uint32_t i = KNZ_TC_STATIC + (uint32_t)lane;
        d.eh[i] = 0; d.ed[i] = (1u << 24) | i; d.ep[i] = KNZ_TC_PTR_ESCAPE | (lane == 0 ? 0x0Eu : 0x0Fu);
        atomicMax(&d.map[0], (int32_t)i);
    }
    wave_sync();
}

__device__ __forceinline__ int knz_tc_emit_index1(uint8_t* dst, int val) {           // :936-953
    if (val < 128) { dst[0] = (uint8_t)val; return 1; }
    if (val < 16384) { dst[0] = (uint8_t)(0x80 | (val >> 7)); dst[1] = (uint8_t)(val & 0x7F); return 2; }
    dst[0] = (uint8_t)(0xE0 | (val >> 14)); dst[1] = (uint8_t)(0x80 | (val >> 7)); dst[2] = (uint8_t)(val & 0x7F);
    return 3;
}
__device__ __forceinline__ int knz_tc_emit_index2(uint8_t* dst, int w) {             // :1489-1511
    w++;
    if (w >= 64) {
        if (w >= 8192) { dst[0] = (uint8_t)(0xF0 | (w >> 16)); dst[1] = (uint8_t)(w >> 8); dst[2] = (uint8_t)w; return 3; }
        dst[0] = (uint8_t)(0xC0 | (w >> 8)); dst[1] = (uint8_t)w;
        return 2;
    }
    dst[0] = (uint8_t)(0x80 | w);
    return 1;
}
// emitSymbols (:884-934 / :1415-1487): literal bytes with the format's escapes; returns the bytes written or room + 1 when they do not fit
__device__ __forceinline__ int knz_tc_emit_symbols(const uint8_t* src, int len, uint8_t* dst, int room, uint32_t kind, bool crlf, int staticSize) {
    int o = 0;
    if (kind == 1) {
        for (int i = 0; i < len; i++) {
            const uint32_t cur = src[i];
            if (o >= room) return room + 1;
            if (cur == 0x0F || cur == 0x0E) {
                dst[o++] = 0x0F;
                const int idx = cur == 0x0F ? staticSize - 1 : staticSize - 2;
                const int lenIdx = idx >= 16384 ? 3 : (idx < 128 ? 1 : 2);
                if (o + lenIdx >= room) return room + 1;
                o += knz_tc_emit_index1(dst + o, idx);
            } else if (cur == 0x0D) { if (!crlf) dst[o++] = (uint8_t)cur; }
            else dst[o++] = (uint8_t)cur;
        }
        return o;
    }
    for (int i = 0; i < len; i++) {
        const uint32_t cur = src[i];
        if (cur == 0x0F) { if (o + 1 >= room) return room + 1; dst[o++] = 0x0F; dst[o++] = 0x0F; }
        else if (cur == 0x0D) { if (!crlf) { if (o >= room) return room + 1; dst[o++] = (uint8_t)cur; } }
        else {
            if (cur >= 0x80) { if (o >= room) return room + 1; dst[o++] = 0x0F; }
            if (o >= room) return room + 1;
            dst[o++] = (uint8_t)cur;
        }
    }
    return o;
}

// Forward scan (:692-867 / :1225-1398), one lane per block after the wave has filled the static part of the dictionary.
__global__ __launch_bounds__(64) void knz_text_forward_chain_kernel(TextArgs a) {
    const uint32_t b = blockIdx.x;
    const int lane = threadIdx.x;
    if (!a.active[b]) return;
    const int mode = a.tmode[b];
    if (mode == -3) return;                                              // done by the parallel kernel (text_par.hip)
    if (mode < 0) { if (lane == 0) { a.ok[b] = mode == -2 ? 1 : 0; a.out_len[b] = 0; } return; }
    const int count = (int)a.in_len[b];
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    TextDict d;
    knz_tc_dict_reset(d, a, b, count, src, lane);
    if (lane != 0) return;
    atomicAdd(a.chain_count, 1u);
    const uint32_t kind = a.kind;
    const bool crlf = (mode & 0x40) != 0;
    const int srcEnd = count, dstEnd = count, dstEndRef = kind == 1 ? dstEnd - 4 : dstEnd - 3;
    int emitAnchor = 0, srcIdx = 0, dstIdx = 1;
    dst[0] = (uint8_t)mode;
    while (srcIdx < srcEnd && src[srcIdx] == ' ') { dst[dstIdx++] = ' '; srcIdx++; emitAnchor++; }
    if (srcIdx >= srcEnd) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; a.out_len[b] = 0; return; }      // (index out of range in the reference: a block of spaces never gets here)
    int delimAnchor = knz_tc_is_text(src[srcIdx]) ? srcIdx - 1 : srcIdx;
    bool failed = false;
    uint32_t h1 = KNZ_TC_HASH1, h2 = KNZ_TC_HASH1;                       // hashes of the letters since delimAnchor (h2: first letter's case flipped)
    while (srcIdx < srcEnd) {
        const uint32_t cur = src[srcIdx];
        if (knz_tc_is_text(cur)) {
            if (srcIdx == delimAnchor + 1) { h1 = knz_tc_hash_step(KNZ_TC_HASH1, cur); h2 = knz_tc_hash_step(KNZ_TC_HASH1, cur ^ 0x20); }
            else { const uint32_t h = cur * KNZ_TC_HASH2; h1 = (h1 * KNZ_TC_HASH1) ^ h; h2 = (h2 * KNZ_TC_HASH1) ^ h; }
            srcIdx++;
            continue;
        }
        if (srcIdx > delimAnchor + 2 && knz_tc_is_delim(cur)) {
            const int length = srcIdx - delimAnchor - 1;
            if (length <= 31) {
                const uint8_t* w = src + delimAnchor + 1;
                const int pe1 = d.map[h1 & d.mask];
                int pe = -1;
                if (pe1 >= 0 && d.eh[pe1] == h1 && (int)(d.ed[pe1] >> 24) == length) pe = pe1;
                else { const int pe2 = d.map[h2 & d.mask]; if (pe2 >= 0 && d.eh[pe2] == h2 && (int)(d.ed[pe2] >> 24) == length) pe = pe2; }
                if (pe >= 0 && !d.same_tail(d.ep[pe], w, length)) pe = -1;
                if (pe < 0) {
                    if ((length > 3 || (length == 3 && d.words < 16384)) && pe1 < 0) d.insert(h1, length, (uint32_t)(delimAnchor + 1));
                } else {
                    if (emitAnchor != delimAnchor || src[delimAnchor] != ' ')
                        dstIdx += knz_tc_emit_symbols(src + emitAnchor, delimAnchor + 1 - emitAnchor, dst + dstIdx, dstEnd - dstIdx, kind, crlf, d.staticSize);
                    if (dstIdx >= dstEndRef) { failed = true; break; }
                    const int idx = (int)(d.ed[pe] & 0x7FFFFu);
                    if (kind == 1) { dst[dstIdx++] = pe == pe1 ? 0x0F : 0x0E; dstIdx += knz_tc_emit_index1(dst + dstIdx, idx); }
                    else { if (pe != pe1) dst[dstIdx++] = 0x80; dstIdx += knz_tc_emit_index2(dst + dstIdx, idx); }
                    emitAnchor = delimAnchor + 1 + (int)(d.ed[pe] >> 24);
                }
            }
        }
        delimAnchor = srcIdx;
        srcIdx++;
    }
    if (!failed) {
        dstIdx += knz_tc_emit_symbols(src + emitAnchor, srcEnd - emitAnchor, dst + dstIdx, dstEnd - dstIdx, kind, crlf, d.staticSize);
        if (dstIdx > dstEnd) failed = true;
    }
    a.ok[b] = failed ? 0 : 1;
    a.out_len[b] = failed ? 0u : (uint32_t)dstIdx;
}

// Inverse scan (:955-1116 / :1513-1718), one lane per block. Dynamic entries point into the encoded block (literal words).
__global__ __launch_bounds__(64) void knz_text_inverse_chain_kernel(TextArgs a) {
    const uint32_t b = blockIdx.x;
    const int lane = threadIdx.x;
    if (!a.active[b] || a.tmode[b] == -3) return;                        // (-3: done by the parallel kernel, text_par.hip)
    const int64_t srcEnd = (int64_t)a.in_len[b], dstEnd = (int64_t)a.out_cap;
    const uint8_t* src = (const uint8_t*)a.in_ptr[b];
    uint8_t* dst = (uint8_t*)a.out_ptr[b];
    if (srcEnd == 0 || dstEnd == 0) { if (lane == 0) { a.ok[b] = 1; a.out_len[b] = 0; } return; }
    if (srcEnd < 2 || srcEnd > (1 << 30)) { if (lane == 0) { a.ok[b] = -KNZ_ERR_PROCESS_BLOCK; a.out_len[b] = 0; } return; }
    TextDict d;
    knz_tc_dict_reset(d, a, b, (int)(dstEnd > (1 << 30) ? (1 << 30) : dstEnd), src, lane);
    if (lane != 0) return;
    atomicAdd(a.chain_count, 1u);
    const uint32_t kind = a.kind;
    const bool crlf = (src[0] & 0x40) != 0;
    bool wordRun = false, bad = false;
    int64_t srcIdx = 1, dstIdx = 0;
    int64_t delimAnchor = knz_tc_is_text(src[srcIdx]) ? srcIdx - 1 : srcIdx;
    uint32_t h1 = KNZ_TC_HASH1;
    while (srcIdx < srcEnd && dstIdx < dstEnd) {
        uint32_t cur = src[srcIdx];
        if (knz_tc_is_text(cur)) {
            h1 = knz_tc_hash_step(srcIdx == delimAnchor + 1 ? KNZ_TC_HASH1 : h1, cur);
            dst[dstIdx++] = (uint8_t)cur; srcIdx++;
            continue;
        }
        if (srcIdx > delimAnchor + 3 && knz_tc_is_delim(cur)) {
            const int length = (int)(srcIdx - delimAnchor - 1);
            if (length <= 31) {
                const uint8_t* w = src + delimAnchor + 1;
                const int pe1 = d.map[h1 & d.mask];
                const bool found = pe1 >= 0 && d.eh[pe1] == h1 && (int)(d.ed[pe1] >> 24) == length && d.same_tail(d.ep[pe1], w, length);
                if (!found && (length > 3 || d.words < 16384) && pe1 < 0) d.insert(h1, length, (uint32_t)(delimAnchor + 1));
            }
        }
        srcIdx++;
        const bool isRef = kind == 1 ? (cur == 0x0F || cur == 0x0E) : cur >= 128;
        if (isRef) {
            int idx;
            uint32_t flip = 0;
            if (kind == 1) {
                if (cur == 0x0E) flip = 0x20;
                if (srcIdx >= srcEnd) { bad = true; break; }
                idx = src[srcIdx++];
                if (idx >= 128) {
                    idx &= 0x7F;
                    if (srcIdx >= srcEnd) { bad = true; break; }
                    int idx2 = src[srcIdx++];
                    if (idx2 >= 0x80) {
                        idx = ((idx & 0x1F) << 7) | (idx2 & 0x7F);
                        if (srcIdx >= srcEnd) { bad = true; break; }
                        idx2 = src[srcIdx++];
                    }
                    idx = (idx << 7) | idx2;
                    if (idx >= d.dictSize) { bad = true; break; }
                }
            } else {
                if (cur == 0x80) { flip = 0x20; if (srcIdx >= srcEnd) { bad = true; break; } cur = src[srcIdx++]; }
                idx = (int)(cur & 0x7F);
                if (idx >= 64) {
                    if (idx >= 112) {
                        if (srcIdx + 1 >= srcEnd) { bad = true; break; }
                        idx = ((idx & 0x0F) << 16) | ((int)src[srcIdx] << 8) | (int)src[srcIdx + 1];
                        srcIdx += 2;
                    } else {
                        if (srcIdx >= srcEnd) { bad = true; break; }
                        idx = ((idx & 0x1F) << 8) | (int)src[srcIdx];
                        srcIdx++;
                    }
                    if (idx > d.dictSize) { bad = true; break; }
                } else if (idx == 0) { bad = true; break; }
                idx--;
            }
            // entries that were never filled have no text (pe.ptr == nil); before the dictionary wraps those are the ones from `words` on
            const bool filled = idx < d.staticSize || d.wrapped || idx < d.words;
            if (idx < 0 || idx >= d.dictSize) { bad = true; break; }
            const int length = filled ? (int)(d.ed[idx] >> 24) & 0xFF : 0;
            if (length > 1) {
                if (wordRun) { if (dstIdx >= dstEnd) { bad = true; break; } dst[dstIdx++] = ' '; }
                wordRun = true;
                delimAnchor = srcIdx;
            } else { wordRun = false; delimAnchor = srcIdx - 1; }
            if (!filled || dstIdx + length >= dstEnd) { bad = true; break; }
            const uint32_t ptr = d.ep[idx];
            for (int k = 0; k < length; k++) dst[dstIdx + k] = (uint8_t)d.ent_byte(ptr, k);
            dst[dstIdx] ^= (uint8_t)flip;
            dstIdx += length;
        } else {
            if (kind == 2 && cur == 0x0F) {
                if (srcIdx >= srcEnd || dstIdx >= dstEnd) { bad = true; break; }
                dst[dstIdx++] = src[srcIdx++];
            } else {
                if (crlf && cur == 0x0A) {
                    dst[dstIdx++] = 0x0D;
                    if (dstIdx >= dstEnd) { bad = true; break; }
                }
                dst[dstIdx++] = (uint8_t)cur;
            }
            wordRun = false;
            delimAnchor = srcIdx - 1;
        }
    }
    if (!bad && srcIdx != srcEnd) bad = true;
    a.ok[b] = bad ? -KNZ_ERR_PROCESS_BLOCK : 1;
    a.out_len[b] = bad ? 0u : (uint32_t)dstIdx;
}
