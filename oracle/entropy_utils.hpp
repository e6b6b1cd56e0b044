// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// CPU restatement of kanzi-go entropy helpers.
//   v2/entropy/EntropyUtils.go:38-67   EncodeAlphabet
//   v2/entropy/EntropyUtils.go:71-119  DecodeAlphabet
//   v2/entropy/EntropyUtils.go:123-260 NormalizeFrequencies
//   v2/entropy/EntropyUtils.go:264-296 WriteVarInt / ReadVarInt
//   v2/entropy/ExpGolombCodec.go:104-112,159-190 signed Exp-Golomb
//   v2/internal/Global.go:156-172 Log2NoCheck ; :220-344 ComputeHistogram
#pragma once
#include "bitstream.hpp"
#include <algorithm>

namespace knzo {

// Global.go:156-172 ; x >= 1
static inline uint32_t log2NoCheck(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x); }

// Global.go:220-251 order 0 (no total)
static inline void histogramO0(const uint8_t* block, size_t n, int64_t* freqs /*256*/) {
    for (size_t i = 0; i < n; i++) freqs[block[i]]++;
}

// Global.go:252-299 order 1 with totals (stride 257): first symbol has context 0, every other
// symbol has its predecessor as context (the 4-way split in the reference is an ILP device:
// prv1..3 are initialised with the byte preceding each quarter).
static inline void histogramO1Total(const uint8_t* block, size_t n, int64_t* freqs /*256*257*/) {
    size_t prv = 0;
    for (size_t i = 0; i < n; i++) {
        freqs[prv + block[i]]++;
        freqs[prv + 256]++;
        prv = 257 * (size_t)block[i];
    }
}

// EntropyUtils.go:38-67
static inline int encodeAlphabet(BitWriter& obs, const int* alphabet, int count) {
    if (count > 256) throw KnzError(ERR_PROCESS_BLOCK, "The max alphabet length is 256");
    if (count == 0) {
        obs.writeBit(0); // _FULL_ALPHABET
        obs.writeBit(1); // _ALPHABET_0
    } else if (count == 256) {
        obs.writeBit(0);
        obs.writeBit(0); // _ALPHABET_256
    } else {
        obs.writeBit(1); // _PARTIAL_ALPHABET
        uint8_t masks[32] = {0};
        for (int i = 0; i < count; i++) masks[alphabet[i] >> 3] |= (uint8_t)(1u << (alphabet[i] & 7));
        int lastMask = alphabet[count - 1] >> 3;
        obs.writeBits((uint64_t)lastMask, 5);
        obs.writeArray(masks, 8u * (unsigned)(lastMask + 1));
    }
    return count;
}

// EntropyUtils.go:71-119 ; alphabetCap = len(alphabet)
static inline int decodeAlphabet(BitReader& ibs, int* alphabet, int alphabetCap) {
    if (ibs.readBit() == 0) {
        if (ibs.readBit() == 1) return 0;
        if (256 > alphabetCap) throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: incorrect alphabet size");
        for (int i = 0; i < 256; i++) alphabet[i] = i;
        return 256;
    }
    int lastMask = (int)ibs.readBits(5);
    uint8_t masks[32] = {0};
    int count = 0;
    ibs.readArray(masks, 8u * (unsigned)(lastMask + 1));
    for (int i = 0; i <= lastMask; i++) {
        int n = i * 8;
        for (int j = 0; j < 8; j++) {
            if (((masks[i] >> j) & 1) == 0) continue;
            if (count >= alphabetCap) throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: incorrect alphabet size");
            alphabet[count++] = n + j;
        }
    }
    return count;
}

// EntropyUtils.go:123-260. freqs has freqsLen entries, alphabet has alphabetLen entries.
// Mutates freqs in place while selecting idxMax, exactly like the reference.
static inline int normalizeFrequencies(int64_t* freqs, int freqsLen, int* alphabet, int alphabetLen,
                                       int64_t totalFreq, int64_t scale) {
    if (alphabetLen > 256) throw KnzError(ERR_PROCESS_BLOCK, "Invalid alphabet size parameter");
    if (scale < 256 || scale > 65536) throw KnzError(ERR_PROCESS_BLOCK, "Invalid range parameter");
    if (alphabetLen == 0 || totalFreq == 0) return 0;
    int alphabetSize = 0;

    if (totalFreq == scale) { // :143-152
        for (int i = 0; i < 256; i++) {
            if (i >= freqsLen) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); // Go slice bound panic
            if (freqs[i] != 0) {
                if (alphabetSize >= alphabetLen) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
                alphabet[alphabetSize++] = i;
            }
        }
        return alphabetSize;
    }

    int64_t sumScaledFreq = 0, sumFreq = 0;
    int idxMax = 0;

    for (int i = 0; i < alphabetLen; i++) { // :159-191
        alphabet[i] = 0;
        int64_t f = freqs[i];
        if (f == 0) continue;
        int64_t sf = freqs[i] * scale;
        int64_t scaledFreq;
        if (sf <= totalFreq) scaledFreq = 1;
        else scaledFreq = (sf + (totalFreq >> 1)) / totalFreq;
        alphabet[alphabetSize++] = i;
        sumScaledFreq += scaledFreq;
        freqs[i] = scaledFreq;
        sumFreq += f;
        if (scaledFreq > freqs[idxMax]) idxMax = i;
        if (sumFreq >= totalFreq) break;
    }

    if (alphabetSize == 0) return 0;
    if (alphabetSize == 1) { freqs[alphabet[0]] = scale; return 1; }
    if (sumScaledFreq == scale) return alphabetSize;

    int64_t delta = sumScaledFreq - scale;
    int64_t errThr = freqs[idxMax] >> 4;
    int64_t inc;
    int64_t absDelta = delta < 0 ? -delta : delta;

    if (absDelta <= errThr) { // :214-218
        freqs[idxMax] -= delta;
        return alphabetSize;
    }

    if (delta < 0) {
        delta += errThr;
        freqs[idxMax] += errThr;
        inc = 1;
        delta = -delta;
    } else {
        delta -= errThr;
        freqs[idxMax] -= errThr;
        inc = -1;
    }

    int round = 1;
    while (round < 6 && delta > 0) { // :233-257
        int adjustments = 0;
        round++;
        for (int k = 0; k < alphabetSize; k++) {
            int idx = alphabet[k];
            if (freqs[idx] <= 2) continue;
            freqs[idx] += inc;
            adjustments++;
            delta--;
            if (delta == 0) break;
        }
        if (adjustments == 0) break;
    }

    freqs[idxMax] = std::max<int64_t>(freqs[idxMax] - delta, 1);
    return alphabetSize;
}

// EntropyUtils.go:264-275
static inline int writeVarInt(BitWriter& bs, uint32_t value) {
    int res = 1;
    while (value >= 128) {
        bs.writeBits((uint64_t)(0x80 | (value & 0x7F)), 8);
        value >>= 7;
        res++;
    }
    bs.writeBits((uint64_t)value, 8);
    return res;
}

// EntropyUtils.go:278-296
static inline uint32_t readVarInt(BitReader& bs) {
    uint32_t res = 0;
    unsigned shift = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t value = (uint32_t)bs.readBits(8);
        res |= (value & 0x7F) << shift;
        if (value < 128) return res;
        shift += 7;
    }
    uint32_t value = (uint32_t)bs.readBits(8);
    return res | ((value & 0x0F) << 28);
}

// Signed Exp-Golomb emit word = (length<<9)|bits, closed form of the table at
// ExpGolombCodec.go:45-62 (checked entry-by-entry in tests/test_oracle_units.py):
// v as int8, n=|v|+1, L=floor(log2 n): (n<<1|sign) in 2L+2 bits.
static inline uint32_t expGolombSignedWord(uint8_t val) {
    int v = (int8_t)val;
    uint32_t sign = v < 0 ? 1u : 0u;
    uint32_t n = (uint32_t)(v < 0 ? -v : v) + 1;
    uint32_t L = log2NoCheck(n);
    uint32_t len = 2 * L + 2;
    uint32_t bits = (n << 1) | sign;
    return (len << 9) | (bits & 0x1FF);
}

// ExpGolombCodec.go:104-112 (signed cache)
static inline void expGolombEncodeByte(BitWriter& bs, uint8_t val) {
    if (val == 0) { bs.writeBit(1); return; }
    uint32_t emit = expGolombSignedWord(val);
    bs.writeBits((uint64_t)(emit & 0x1FF), emit >> 9);
}

// ExpGolombCodec.go:159-190 (signed)
static inline uint8_t expGolombDecodeByte(BitReader& bs) {
    if (bs.readBit() == 1) return 0;
    unsigned lg = 1;
    for (;;) {
        if (bs.readBit() == 1) break;
        lg++;
    }
    lg &= 7;
    uint64_t val = bs.readBits(lg + 1);
    uint64_t res = (val >> 1) + ((uint64_t)1 << lg) - 1;
    if (val & 1) res = ~res + 1;
    return (uint8_t)res;
}

} // namespace knzo
