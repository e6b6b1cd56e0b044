// TEST INFRASTRUCTURE ONLY (see oracle/README.md): CPU restatement of kanzi-go's TEXT transform (v2/transform/TextCodec.go), the dictionary
// word-replacement stage in front of UTF / BWT in the `-l 5..9` presets. Two stream formats live behind one transform id: "codec 1" (escape
// tokens 0x0F / 0x0E + a 1..3 byte index; used in front of the bit-wise entropy coders) and "codec 2" (indexes as bytes >= 0x80; picked by
// Factory.go:100-120 when the entropy stage is NONE / ANS0 / HUFFMAN / RANGE). Both build the same dynamic dictionary while they scan.
// Pinned like the rest of the oracle by oracle/_ref (the reference's own TextCodec.go, translated mechanically: tests/test_ref_build.py); the static dictionary and every constant come from the fixture
// tests/golden/reference_constants.json (extracted from the reference source by tests/golden/make_golden.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace knzo {

static inline uint32_t getMagicType(const uint8_t* src, size_t n);   // stream.hpp (internal/Magic.go)

// internal/Global.go:26-40
enum : int { DT_UNDEFINED = 0, DT_TEXT = 1, DT_MULTIMEDIA = 2, DT_EXE = 3, DT_NUMERIC = 4, DT_BASE64 = 5, DT_DNA = 6, DT_BIN = 7, DT_UTF8 = 8,
             DT_SMALL_ALPHABET = 9 };
// the task's ctx map travels in thread-locals: ctx["dataType"], ctx["blockSize"], ctx["entropy"] (the codec type id)
static thread_local int tlsDataType = DT_UNDEFINED;
static thread_local uint32_t tlsBlockSize = 0;        // 0 = key absent
static thread_local uint32_t tlsEntropyType = 0xFFFFFFFFu;   // 0xFFFFFFFF = key absent

enum : int { TC_THRESHOLD1 = 128, TC_THRESHOLD2 = 128 * 128, TC_THRESHOLD3 = 64, TC_THRESHOLD4 = 64 * 128, TC_MAX_DICT_SIZE = 1 << 19,
             TC_MAX_WORD_LENGTH = 31, TC_MIN_BLOCK_SIZE = 1024, TC_MAX_BLOCK_SIZE = 1 << 30, TC_ESCAPE_TOKEN1 = 0x0F, TC_ESCAPE_TOKEN2 = 0x0E,
             TC_MASK_FLIP_CASE = 0x80, TC_MASK_NOT_TEXT = 0x80, TC_MASK_CRLF = 0x40, TC_MASK_XML_HTML = 0x20, TC_MASK_DT = 0x0F,
             TC_MASK_LENGTH = 0x0007FFFF, TC_STATIC_WORDS = 1024 };
static const uint32_t TC_HASH1 = 0x7FEB352Du;          // int32 arithmetic in the reference; the same bits in uint32
static const uint32_t TC_HASH2 = 0x846CA68Bu;

static inline bool tcIsText(uint8_t v) { v |= 0x20; return v >= 'a' && v <= 'z'; }            // :492-494
static inline bool tcIsDelimiter(uint8_t v) {                                                  // :409-448
    if (v >= ' ' && v <= '/') return true;
    if (v >= ':' && v <= '?') return true;
    switch (v) { case '\n': case '\r': case '\t': case '_': case '|': case '{': case '}': case '[': case ']': return true; default: return false; }
}
static inline uint32_t tcHashStep(uint32_t h, uint8_t c) { return (h * TC_HASH1) ^ ((uint32_t)c * TC_HASH2); }

struct TextDictEntry {
    uint32_t hash = 0;          // full word hash
    int32_t data = 0;           // length << 24 | index
    const uint8_t* ptr = nullptr;
};

// :451-490 createDictionary over the 1024-word list: words start at their upper case letter, are stored lower case
struct TextStaticDict {
    std::vector<uint8_t> letters;
    TextDictEntry e[TC_STATIC_WORDS];
    int words = 0;
    TextStaticDict() {
        static const char src[] =
#include "text_dict.inc"
            ;
        letters.assign(src, src + sizeof(src) - 1);
        int anchor = 0;
        uint32_t h = TC_HASH1;
        const int n = (int)letters.size();
        for (int i = 0; i < n && words < TC_STATIC_WORDS; i++) {
            if (letters[i] >= 'A' && letters[i] <= 'Z') {
                if (i > anchor) {
                    e[words].ptr = letters.data() + anchor; e[words].hash = h; e[words].data = ((i - anchor) << 24) | words;
                    words++; anchor = i; h = TC_HASH1;
                }
                letters[i] ^= 0x20;
            }
            h = tcHashStep(h, letters[i]);
        }
        if (words < TC_STATIC_WORDS) {
            e[words].ptr = letters.data() + anchor; e[words].hash = h; e[words].data = ((n - anchor) << 24) | words;
            words++;
        }
    }
};
static inline const TextStaticDict& textStaticDict() { static const TextStaticDict d; return d; }

// internal/Global.go:346-420 DetectSimpleType
static inline int detectSimpleType(int count, const int* freqs0) {
    if (count == 0) return DT_UNDEFINED;
    int sum = 0;
    for (const char* p = "acgntuACGNTU"; *p; p++) sum += freqs0[(uint8_t)*p];
    if (sum > count - count / 12) return DT_DNA;
    sum = 0;
    for (const char* p = "0123456789+-*/=,.:; "; *p; p++) sum += freqs0[(uint8_t)*p];
    if (sum == count) return DT_NUMERIC;
    sum = 0;
    for (const char* p = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/"; *p; p++) sum += freqs0[(uint8_t)*p];
    if (sum + freqs0[0x3D] == count) return DT_BASE64;
    sum = 0;
    for (int i = 0; i < 256; i++) if (freqs0[i] > 0) sum++;
    if (sum == 256) return DT_BIN;
    if (sum <= 4) return DT_SMALL_ALPHABET;
    return DT_UNDEFINED;
}

// :308-397: what a block that is not text looks like (simple types, else "all byte pairs are legal UTF-8 starts")
static inline uint8_t tcDetectTextType(const int* freqs0, const int* freqs1 /* [256][256] */, int count) {
    const int dt = detectSimpleType(count, freqs0);
    if (dt != DT_UNDEFINED) return (uint8_t)(TC_MASK_NOT_TEXT | dt);
    int sum = freqs0[0xC0] + freqs0[0xC1];
    for (int i = 0xF5; i < 256; i++) sum += freqs0[i];
    if (sum != 0) return TC_MASK_NOT_TEXT;
    int sum2 = 0;
    auto f = [&](int a, int b) { return freqs1[a * 256 + b]; };
    for (int i = 0; i < 256; i++) {
        if (i < 0xA0 || i > 0xBF) sum += f(0xE0, i);
        if (i < 0x80 || i > 0x9F) sum += f(0xED, i);
        if (i < 0x90 || i > 0xBF) sum += f(0xF0, i);
        if (i < 0x80 || i > 0x8F) sum += f(0xF4, i);
        if (i < 0x80 || i > 0xBF) {
            for (int j = 0xC2; j <= 0xDF; j++) sum += f(j, i);
            for (int j = 0xE1; j <= 0xEC; j++) sum += f(j, i);
            sum += f(0xF1, i) + f(0xF2, i) + f(0xF3, i) + f(0xEE, i) + f(0xEF, i);
        } else {
            sum2 += freqs0[i];
        }
        if (sum != 0) return TC_MASK_NOT_TEXT;
    }
    return sum2 >= count / 8 ? (uint8_t)(TC_MASK_NOT_TEXT | DT_UTF8) : (uint8_t)TC_MASK_NOT_TEXT;
}

// :187-306: the mode byte of a block (not text | CRLF | XML-HTML | data type)
static inline uint8_t tcComputeStats(const uint8_t* block, int count, int* freqs0, std::vector<int>& freqs1, bool strict) {
    if (!strict && getMagicType(block, (size_t)count) != 0) return TC_MASK_NOT_TEXT;
    freqs1.assign(65536, 0);
    uint8_t prv = 0;
    for (int i = 0; i < count; i++) { const uint8_t cur = block[i]; freqs0[cur]++; freqs1[prv * 256 + cur]++; prv = cur; }
    int nbTextChars = freqs0[0x0D] + freqs0[0x0A];
    int nbASCII = 0;
    for (int i = 0; i < 128; i++) { if (tcIsText((uint8_t)i)) nbTextChars += freqs0[i]; nbASCII += freqs0[i]; }
    const int nbBinChars = count - nbASCII;
    bool notText;
    if (nbBinChars > (count >> 2)) notText = true;
    else {
        notText = nbTextChars < count / 4;
        if (strict) notText = notText || freqs0[0] >= count / 100 || (nbASCII / 95) < (count / 100);
        else notText = notText || freqs0[32] < count / 50;
    }
    if (notText) return tcDetectTextType(freqs0, freqs1.data(), count);
    uint8_t res = 0;
    if (nbBinChars <= count - count / 10) {
        const int f1 = freqs0['<'], f2 = freqs0['>'];
        const int f3 = freqs1['&' * 256 + 'a'] + freqs1['&' * 256 + 'g'] + freqs1['&' * 256 + 'l'] + freqs1['&' * 256 + 'q'];
        int minFreq = (count - nbBinChars) >> 9;
        if (minFreq < 2) minFreq = 2;
        if (f1 >= minFreq && f2 >= minFreq && f3 > 0) {
            if (f1 < f2) { if (f1 >= f2 - f2 / 100) res |= TC_MASK_XML_HTML; }
            else if (f2 < f1) { if (f2 >= f1 - f1 / 100) res |= TC_MASK_XML_HTML; }
            else res |= TC_MASK_XML_HTML;
        }
    }
    if (freqs0[0x0D] != 0 && freqs0[0x0D] == freqs0[0x0A]) {
        bool isCRLF = true;
        for (int i = 0; i < 256; i++) {
            if (i != 0x0A && freqs1[0x0D * 256 + i] != 0) { isCRLF = false; break; }
            if (i != 0x0D && freqs1[i * 256 + 0x0A] != 0) { isCRLF = false; break; }
        }
        if (isCRLF) res |= TC_MASK_CRLF;
    }
    return res;
}

static inline unsigned tcLog2(uint32_t x) { unsigned l = 0; while (x > 1) { x >>= 1; l++; } return l; }

// One object per call, as in the stream path (encodingTask / decodingTask build a new sequence for every block).
class TextCodec {
public:
    // Factory.go:100-120 (ctx["textcodec"]), TextCodec.go:610-650 / :1137-1188 (hash table size from ctx["blockSize"])
    TextCodec() {
        kind = 1;
        if (tlsEntropyType != 0xFFFFFFFFu) {
            const uint32_t e = tlsEntropyType;                        // NONE 0, HUFFMAN 1, RANGE 4, ANS0 5 (entropy/EntropyCodecFactory.go:25-42)
            if (e == 0 || e == 1 || e == 4 || e == 5) kind = 2;
        }
        unsigned lg = 13;
        if (tlsBlockSize != 0) {
            if (kind == 1) { if (tlsBlockSize >= 8) { lg = tcLog2(tlsBlockSize / 8); if (lg > 26) lg = 26; if (lg < 13) lg = 13; } }
            else if (tlsBlockSize >= 32) { lg = tcLog2(tlsBlockSize / 32); if (lg > 24) lg = 24; if (lg < 13) lg = 13; }
        }
        if (tlsEntropyType == 9) lg++;                                // "TPAQX"
        logHashSize = lg;
        hashMask = ((uint32_t)1 << lg) - 1;
    }

    size_t forward(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {
        if (n == 0 || dstCap == 0) return 0;
        if (n < (size_t)TC_MIN_BLOCK_SIZE) throw SkipTransform("The min text transform block size is 1024");
        if (n > (size_t)TC_MAX_BLOCK_SIZE) throw SkipTransform("The max text transform block size is 1 GB");
        const int count = (int)n;
        if (dstCap < n) throw SkipTransform("Output buffer is too small");
        {   // :699-708 / :1232-1241
            const int dt = tlsDataType;
            if (dt != DT_UNDEFINED && dt != DT_TEXT && dt != DT_BIN) throw SkipTransform("Input is not text, skip");
        }
        int freqs0[256] = {0};
        std::vector<int> freqs1;
        const uint8_t mode = tcComputeStats(src, count, freqs0, freqs1, kind == 1);
        if (mode & TC_MASK_NOT_TEXT) { tlsDataType = mode & TC_MASK_DT; throw SkipTransform("Input is not text, skip"); }
        tlsDataType = DT_TEXT;
        reset(count);
        const int srcEnd = count, dstEnd = count;
        const int dstEndRef = kind == 1 ? dstEnd - 4 : dstEnd - 3;
        int emitAnchor = 0;
        words = staticDictSize;
        isCRLF = (mode & TC_MASK_CRLF) != 0;
        dst[0] = mode;
        int srcIdx = 0, dstIdx = 1;
        while (srcIdx < srcEnd && src[srcIdx] == ' ') { dst[dstIdx++] = ' '; srcIdx++; emitAnchor++; }
        if (srcIdx >= srcEnd) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
        int delimAnchor = tcIsText(src[srcIdx]) ? srcIdx - 1 : srcIdx;
        bool failed = false;
        while (srcIdx < srcEnd) {
            const uint8_t cur = src[srcIdx];
            if (tcIsText(cur)) { srcIdx++; continue; }
            if (srcIdx > delimAnchor + 2 && tcIsDelimiter(cur)) {                         // a word of at least 2 letters ends here
                const int length = srcIdx - delimAnchor - 1;
                if (length <= TC_MAX_WORD_LENGTH) {
                    const uint8_t* w = src + delimAnchor + 1;
                    uint32_t h1 = tcHashStep(TC_HASH1, w[0]);
                    uint32_t h2 = tcHashStep(TC_HASH1, (uint8_t)(w[0] ^ 0x20));           // first letter's case flipped
                    for (int i = 1; i < length; i++) { const uint32_t h = (uint32_t)w[i] * TC_HASH2; h1 = (h1 * TC_HASH1) ^ h; h2 = (h2 * TC_HASH1) ^ h; }
                    const int pe1 = dictMap[h1 & hashMask];
                    int pe = -1;
                    if (pe1 >= 0 && dictList[pe1].hash == h1 && (dictList[pe1].data >> 24) == length) pe = pe1;
                    else { const int pe2 = dictMap[h2 & hashMask]; if (pe2 >= 0 && dictList[pe2].hash == h2 && (dictList[pe2].data >> 24) == length) pe = pe2; }
                    if (pe >= 0 && memcmp(dictList[pe].ptr + 1, w + 1, (size_t)length - 1) != 0) pe = -1;     // hash collision
                    if (pe < 0) {
                        if ((length > 3 || (length == 3 && words < TC_THRESHOLD2)) && pe1 < 0) insertWord(w, h1, length);
                    } else {
                        if (emitAnchor != delimAnchor || src[delimAnchor] != ' ')        // a lone space between two references is implied
                            dstIdx += emitSymbols(src + emitAnchor, delimAnchor + 1 - emitAnchor, dst + dstIdx, dstEnd - dstIdx);
                        if (dstIdx >= dstEndRef) { failed = true; break; }
                        const int idx = dictList[pe].data & TC_MASK_LENGTH;
                        if (kind == 1) { dst[dstIdx++] = pe == pe1 ? TC_ESCAPE_TOKEN1 : TC_ESCAPE_TOKEN2; dstIdx += emitWordIndex1(dst + dstIdx, idx); }
                        else { if (pe != pe1) dst[dstIdx++] = TC_MASK_FLIP_CASE; dstIdx += emitWordIndex2(dst + dstIdx, idx); }
                        emitAnchor = delimAnchor + 1 + (dictList[pe].data >> 24);
                    }
                }
            }
            delimAnchor = srcIdx;
            srcIdx++;
        }
        if (!failed) {
            dstIdx += emitSymbols(src + emitAnchor, srcEnd - emitAnchor, dst + dstIdx, dstEnd - dstIdx);
            if (dstIdx > dstEnd) failed = true;
        }
        if (failed) throw SkipTransform("Text transform failed. Output buffer too small");
        return (size_t)dstIdx;
    }

    size_t inverse(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {
        if (n == 0 || dstCap == 0) return 0;
        if (n < 2) throw KnzError(ERR_PROCESS_BLOCK, "Input block is too small");
        if (n > (size_t)TC_MAX_BLOCK_SIZE) throw KnzError(ERR_PROCESS_BLOCK, "The max text transform block size is 1 GB");
        reset((int)std::min<size_t>(dstCap, (size_t)TC_MAX_BLOCK_SIZE));
        const int64_t srcEnd = (int64_t)n, dstEnd = (int64_t)dstCap;
        words = staticDictSize;
        bool wordRun = false;
        isCRLF = (src[0] & TC_MASK_CRLF) != 0;
        int64_t srcIdx = 1, dstIdx = 0;
        int64_t delimAnchor = tcIsText(src[srcIdx]) ? srcIdx - 1 : srcIdx;
        auto rd = [&](int64_t i) -> uint8_t { if (i >= srcEnd) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); return src[i]; };
        auto wr = [&](int64_t i, uint8_t v) { if (i >= dstEnd) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); dst[i] = v; };
        while (srcIdx < srcEnd && dstIdx < dstEnd) {
            uint8_t cur = src[srcIdx];
            if (tcIsText(cur)) { dst[dstIdx++] = cur; srcIdx++; continue; }
            if (srcIdx > delimAnchor + 3 && tcIsDelimiter(cur)) {                        // a literal word of at least 3 letters: same dictionary update as the encoder's
                const int length = (int)(srcIdx - delimAnchor - 1);
                if (length <= TC_MAX_WORD_LENGTH) {
                    const uint8_t* w = src + delimAnchor + 1;
                    uint32_t h1 = TC_HASH1;
                    for (int i = 0; i < length; i++) h1 = tcHashStep(h1, w[i]);
                    const int pe1 = dictMap[h1 & hashMask];
                    const bool found = pe1 >= 0 && dictList[pe1].hash == h1 && (dictList[pe1].data >> 24) == length &&
                                       memcmp(dictList[pe1].ptr + 1, w + 1, (size_t)length - 1) == 0;
                    if (!found && (length > 3 || words < TC_THRESHOLD2) && pe1 < 0) insertWord(w, h1, length);
                }
            }
            srcIdx++;
            const bool isRef = kind == 1 ? (cur == TC_ESCAPE_TOKEN1 || cur == TC_ESCAPE_TOKEN2) : cur >= 128;
            if (isRef) {
                int idx;
                uint8_t flipMask = 0;
                if (kind == 1) {
                    if (cur == TC_ESCAPE_TOKEN2) flipMask = 0x20;
                    idx = rd(srcIdx++);
                    if (idx >= 128) {
                        idx &= 0x7F;
                        int idx2 = rd(srcIdx++);
                        if (idx2 >= 0x80) { idx = ((idx & 0x1F) << 7) | (idx2 & 0x7F); idx2 = rd(srcIdx++); }
                        idx = (idx << 7) | idx2;
                        if (idx >= dictSize) throw KnzError(ERR_PROCESS_BLOCK, "Text transform failed. Invalid index");
                    }
                } else {
                    if (cur == TC_MASK_FLIP_CASE) { flipMask = 0x20; cur = rd(srcIdx++); }
                    idx = cur & 0x7F;
                    if (idx >= 64) {
                        if (idx >= 112) { idx = ((idx & 0x0F) << 16) | ((int)rd(srcIdx) << 8) | rd(srcIdx + 1); srcIdx += 2; }
                        else { idx = ((idx & 0x1F) << 8) | rd(srcIdx); srcIdx++; }
                        if (idx > dictSize) throw KnzError(ERR_PROCESS_BLOCK, "Text transform failed. Invalid index");
                    } else if (idx == 0) throw KnzError(ERR_PROCESS_BLOCK, "Text transform failed. Invalid index");
                    idx--;
                }
                if (idx < 0 || idx >= (int)dictList.size()) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
                const TextDictEntry& pe = dictList[idx];
                const int length = (pe.data >> 24) & 0xFF;
                if (length > 1) {
                    if (wordRun) wr(dstIdx++, ' ');
                    wordRun = true;
                    delimAnchor = srcIdx;
                } else { wordRun = false; delimAnchor = srcIdx - 1; }
                if (pe.ptr == nullptr || dstIdx + length >= dstEnd) throw KnzError(ERR_PROCESS_BLOCK, "Text transform failed. Invalid input data");
                memcpy(dst + dstIdx, pe.ptr, (size_t)length);
                dst[dstIdx] ^= flipMask;
                dstIdx += length;
            } else {
                if (kind == 2 && cur == TC_ESCAPE_TOKEN1) { wr(dstIdx++, rd(srcIdx++)); }
                else {
                    if (isCRLF && cur == 0x0A) {
                        wr(dstIdx++, 0x0D);
                        if (dstIdx >= dstEnd) throw KnzError(ERR_PROCESS_BLOCK, "Text transform failed. Invalid input data");
                    }
                    wr(dstIdx++, cur);
                }
                wordRun = false;
                delimAnchor = srcIdx - 1;
            }
        }
        if (srcIdx != srcEnd) throw KnzError(ERR_PROCESS_BLOCK, "Text transform failed. Source index != expected");
        return (size_t)dstIdx;
    }

    int kind;

private:
    unsigned logHashSize;
    uint32_t hashMask;
    int dictSize = 1 << 13;
    int staticDictSize = TC_STATIC_WORDS;
    int words = 0;
    bool isCRLF = false;
    std::vector<int32_t> dictMap;                  // slot -> entry index, -1 = nil
    std::vector<TextDictEntry> dictList;
    uint8_t esc2 = TC_ESCAPE_TOKEN2, esc1 = TC_ESCAPE_TOKEN1;

    // :652-690 / :1190-1223
    void reset(int count) {
        if (count >= 1024) { unsigned lg = tcLog2((uint32_t)count / 128); if (lg > 18) lg = 18; if (lg < 13) lg = 13; dictSize = 1 << lg; }
        dictMap.assign((size_t)1 << logHashSize, -1);
        dictList.assign((size_t)dictSize, TextDictEntry());
        const TextStaticDict& sd = textStaticDict();
        for (int i = 0; i < TC_STATIC_WORDS; i++) dictList[i] = sd.e[i];
        staticDictSize = TC_STATIC_WORDS;
        if (kind == 1) {                                                        // the two escape bytes as one-letter words
            dictList[TC_STATIC_WORDS].ptr = &esc2; dictList[TC_STATIC_WORDS].hash = 0; dictList[TC_STATIC_WORDS].data = (1 << 24) | TC_STATIC_WORDS;
            dictList[TC_STATIC_WORDS + 1].ptr = &esc1; dictList[TC_STATIC_WORDS + 1].hash = 0; dictList[TC_STATIC_WORDS + 1].data = (1 << 24) | (TC_STATIC_WORDS + 1);
            staticDictSize = TC_STATIC_WORDS + 2;
        }
        for (int i = 0; i < staticDictSize; i++) dictMap[dictList[i].hash & hashMask] = i;
        for (int i = staticDictSize; i < dictSize; i++) { dictList[i].ptr = nullptr; dictList[i].hash = 0; dictList[i].data = i; }
    }

    // :801-822: take the next entry (recycling it once the dictionary has wrapped), point the word's slot at it
    void insertWord(const uint8_t* w, uint32_t h1, int length) {
        TextDictEntry& pe = dictList[words];
        if ((pe.data & TC_MASK_LENGTH) >= staticDictSize) {
            dictMap[pe.hash & hashMask] = -1;           // (a fresh entry has hash 0: slot 0 is cleared by every first use of an entry)
            pe.ptr = w; pe.hash = h1; pe.data = (length << 24) | words;
        }
        dictMap[h1 & hashMask] = words;
        words++;
        if (words >= dictSize) {
            if (dictSize >= TC_MAX_DICT_SIZE) words = staticDictSize;
            else {
                dictList.resize((size_t)dictSize * 2);
                for (int i = dictSize; i < dictSize * 2; i++) { dictList[i].ptr = nullptr; dictList[i].hash = 0; dictList[i].data = i; }
                dictSize <<= 1;
            }
        }
    }

    static int emitWordIndex1(uint8_t* dst, int val) {      // :936-953
        if (val < TC_THRESHOLD1) { dst[0] = (uint8_t)val; return 1; }
        if (val < TC_THRESHOLD2) { dst[0] = (uint8_t)(0x80 | (val >> 7)); dst[1] = (uint8_t)(0x7F & val); return 2; }
        dst[0] = (uint8_t)(0xE0 | (val >> 14)); dst[1] = (uint8_t)(0x80 | (val >> 7)); dst[2] = (uint8_t)(0x7F & val);
        return 3;
    }
    static int emitWordIndex2(uint8_t* dst, int wIdx) {     // :1489-1511
        wIdx++;
        if (wIdx >= TC_THRESHOLD3) {
            if (wIdx >= TC_THRESHOLD4) { dst[0] = (uint8_t)(0xF0 | (wIdx >> 16)); dst[1] = (uint8_t)(wIdx >> 8); dst[2] = (uint8_t)wIdx; return 3; }
            dst[0] = (uint8_t)(0xC0 | (wIdx >> 8)); dst[1] = (uint8_t)wIdx;
            return 2;
        }
        dst[0] = (uint8_t)(0x80 | wIdx);
        return 1;
    }

    // :884-934 / :1415-1487. Returns the bytes written, or dstEnd + 1 when the slice is too small.
    int emitSymbols(const uint8_t* src, int len, uint8_t* dst, int dstEnd) const {
        int dstIdx = 0;
        if (kind == 1) {
            for (int i = 0; i < len; i++) {
                const uint8_t cur = src[i];
                if (dstIdx >= dstEnd) return dstEnd + 1;
                if (cur == TC_ESCAPE_TOKEN1 || cur == TC_ESCAPE_TOKEN2) {
                    dst[dstIdx++] = TC_ESCAPE_TOKEN1;
                    const int idx = cur == TC_ESCAPE_TOKEN1 ? staticDictSize - 1 : staticDictSize - 2;
                    int lenIdx = 2;
                    if (idx >= TC_THRESHOLD2) lenIdx = 3; else if (idx < TC_THRESHOLD1) lenIdx = 1;
                    if (dstIdx + lenIdx >= dstEnd) return dstEnd + 1;
                    dstIdx += emitWordIndex1(dst + dstIdx, idx);
                } else if (cur == 0x0D) { if (!isCRLF) dst[dstIdx++] = cur; }
                else dst[dstIdx++] = cur;
            }
            return dstIdx;
        }
        // codec 2: the unchecked fast path (2*len < dstEnd) writes the same bytes as the checked one, which cannot fail in that case
        for (int i = 0; i < len; i++) {
            const uint8_t cur = src[i];
            if (cur == TC_ESCAPE_TOKEN1) {
                if (dstIdx + 1 >= dstEnd) return dstEnd + 1;
                dst[dstIdx++] = TC_ESCAPE_TOKEN1; dst[dstIdx++] = TC_ESCAPE_TOKEN1;
            } else if (cur == 0x0D) {
                if (!isCRLF) { if (dstIdx >= dstEnd) return dstEnd + 1; dst[dstIdx++] = cur; }
            } else {
                if (cur >= 0x80) { if (dstIdx >= dstEnd) return dstEnd + 1; dst[dstIdx++] = TC_ESCAPE_TOKEN1; }
                if (dstIdx >= dstEnd) return dstEnd + 1;
                dst[dstIdx++] = cur;
            }
        }
        return dstIdx;
    }
};

static inline size_t textForward(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) { TextCodec c; return c.forward(src, n, dst, cap); }
static inline size_t textInverse(const uint8_t* src, size_t n, uint8_t* dst, size_t cap) { TextCodec c; return c.inverse(src, n, dst, cap); }

} // namespace knzo
