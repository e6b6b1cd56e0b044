// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// CPU restatement of kanzi-go's static rANS codec, order 0 and order 1 (bitstream v6).
//   v2/entropy/ANSRangeCodec.go:31-37 constants ; :117-168 constructor (chunk size, log range)
//   :171-214 updateFrequencies ; :216-270 encodeHeader ; :274-311 Write ; :313-329 encodeSymbol
//   :331-405 encodeChunk ; :408-427 rebuildStatistics ; :446-468 encSymbol.reset
//   :605-710 decodeHeader ; :714-757 Read ; :846-858 decodeSymbol ; :860-957 decodeChunkV2
#pragma once
#include "entropy_utils.hpp"

namespace knzo {

static const int64_t ANS_TOP = 1 << 15;
static const int ANS0_CHUNK_SIZE = 16384;
static const int ANS_MAX_CHUNK_SIZE = 1 << 27;

struct AnsEncSymbol {
    int64_t xMax, bias, cmplFreq;
    uint8_t invShift;
    uint64_t invFreq;
    // :446-468
    void reset(int64_t cumFreq, int64_t freq, unsigned logRange) {
        freq = std::min<int64_t>(freq, ((int64_t)1 << logRange) - 1);
        xMax = ((ANS_TOP >> logRange) << 16) * freq;
        cmplFreq = ((int64_t)1 << logRange) - freq;
        if (freq < 2) {
            invFreq = 0xFFFFFFFFull;
            invShift = 32;
            bias = cumFreq + ((int64_t)1 << logRange) - 1;
        } else {
            unsigned shift = 0;
            while (freq > ((int64_t)1 << shift)) shift++;
            invFreq = ((((uint64_t)1 << (shift + 31)) + (uint64_t)(freq - 1)) / (uint64_t)freq) & 0xFFFFFFFFull;
            invShift = (uint8_t)(32 + shift - 1);
            bias = cumFreq;
        }
    }
};

struct AnsEncoder {
    BitWriter& bs;
    unsigned order;
    unsigned logRange;
    int chunkSize;
    std::vector<int64_t> freqs;        // dim*257
    std::vector<AnsEncSymbol> symbols; // dim*256
    std::vector<uint8_t> buffer;

    // :117-168 with args (order) only: chunk 16384 (<<8 for order 1, capped), logRange 12-order
    AnsEncoder(BitWriter& b, unsigned ord) : bs(b), order(ord) {
        int chk = ANS0_CHUNK_SIZE;
        if (order == 1) chk = std::min(chk << 8, ANS_MAX_CHUNK_SIZE);
        int dim = (int)(255 * order + 1);
        freqs.assign((size_t)dim * 257, 0);
        symbols.resize((size_t)dim * 256);
        logRange = std::max<unsigned>(12 - order, 8);
        chunkSize = chk;
    }

    // :216-270
    void encodeHeader(const int* alphabet, int alphabetSize, const int64_t* frequencies, unsigned lr) {
        encodeAlphabet(bs, alphabet, alphabetSize);
        if (alphabetSize <= 1) return;
        int chkSize = alphabetSize < 64 ? 6 : 8;
        unsigned llr = 3;
        while (((unsigned)1 << llr) <= lr) llr++;
        for (int i = 1; i < alphabetSize; i += chkSize) {
            int64_t mx = frequencies[alphabet[i]] - 1;
            unsigned logMax = 0;
            int endj = std::min(i + chkSize, alphabetSize);
            for (int j = i + 1; j < endj; j++)
                if (frequencies[alphabet[j]] - 1 > mx) mx = frequencies[alphabet[j]] - 1;
            while (((int64_t)1 << logMax) <= mx) logMax++;
            bs.writeBits((uint64_t)logMax, llr);
            if (logMax == 0) continue;
            for (int j = i; j < endj; j++) bs.writeBits((uint64_t)(frequencies[alphabet[j]] - 1), logMax);
        }
    }

    // :171-214
    int updateFrequencies(unsigned lr) {
        int res = 0;
        int endk = (int)(255 * order + 1);
        bs.writeBits((uint64_t)(lr - 8), 3);
        int alphabet[256];
        for (int k = 0; k < endk; k++) {
            int64_t* f = &freqs[(size_t)257 * k];
            AnsEncSymbol* symb = &symbols[(size_t)k << 8];
            int alphabetSize = normalizeFrequencies(f, 256, alphabet, 256, f[256], (int64_t)1 << lr);
            if (alphabetSize > 0) {
                int64_t sum = 0;
                for (int i = 0, count = 0; i < 256; i++) {
                    if (f[i] == 0) continue;
                    symb[i].reset(sum, f[i], lr);
                    sum += f[i];
                    count++;
                    if (count >= alphabetSize) break;
                }
            }
            encodeHeader(alphabet, alphabetSize, f, lr);
            res += alphabetSize;
        }
        return res;
    }

    // :408-427
    int rebuildStatistics(const uint8_t* block, size_t len, unsigned lr) {
        std::fill(freqs.begin(), freqs.end(), 0);
        if (order == 0) {
            histogramO0(block, len, freqs.data());
            freqs[256] = (int64_t)len;
        } else {
            size_t quarter = len >> 2;
            if (quarter == 0) {
                histogramO1Total(block, len, freqs.data());
            } else {
                histogramO1Total(block + 0 * quarter, quarter, freqs.data());
                histogramO1Total(block + 1 * quarter, quarter, freqs.data());
                histogramO1Total(block + 2 * quarter, quarter, freqs.data());
                histogramO1Total(block + 3 * quarter, quarter, freqs.data());
            }
        }
        return updateFrequencies(lr);
    }

    // :313-329
    inline void encodeSymbol(int64_t& n, int64_t& st, const AnsEncSymbol& sym) {
        int64_t x = (st >= sym.xMax) ? 1 : 0;
        if (n < x) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); // Go slice bound panic
        buffer[(size_t)n] = (uint8_t)st;
        n -= x;
        buffer[(size_t)n] = (uint8_t)(st >> 8);
        n -= x;
        st >>= (-x & 16);
        st = st + sym.bias + (int64_t)(((uint64_t)st * sym.invFreq) >> sym.invShift) * sym.cmplFreq;
    }

    // :331-405
    void encodeChunk(const uint8_t* block, int64_t len) {
        int64_t st0 = ANS_TOP, st1 = ANS_TOP, st2 = ANS_TOP, st3 = ANS_TOP;
        int64_t n = (int64_t)buffer.size() - 1;
        int64_t end4 = len & -4;
        for (int64_t i = len - 1; i >= end4; i--) { buffer[(size_t)n] = block[i]; n--; }

        if (order == 0) {
            const AnsEncSymbol* symb = symbols.data();
            for (int64_t i = end4 - 1; i > 0; i -= 4) {
                encodeSymbol(n, st0, symb[block[i]]);
                encodeSymbol(n, st1, symb[block[i - 1]]);
                encodeSymbol(n, st2, symb[block[i - 2]]);
                encodeSymbol(n, st3, symb[block[i - 3]]);
            }
        } else if (len > 1) {
            int64_t quarter = end4 >> 2;
            int64_t i0 = 1 * quarter - 2, i1 = 2 * quarter - 2, i2 = 3 * quarter - 2, i3 = end4 - 2;
            // i0+1 == -1 when quarter == 0 : Go panics with index out of range (SURVEY §8c edge case)
            if (quarter == 0) throw KnzError(ERR_PROCESS_BLOCK, "index out of range [-1]");
            int prv0 = block[i0 + 1], prv1 = block[i1 + 1], prv2 = block[i2 + 1], prv3 = block[i3 + 1];
            while (i0 >= 0) {
                int cur0 = block[i0];
                encodeSymbol(n, st0, symbols[(size_t)((cur0 << 8) | prv0)]);
                int cur1 = block[i1];
                encodeSymbol(n, st1, symbols[(size_t)((cur1 << 8) | prv1)]);
                int cur2 = block[i2];
                encodeSymbol(n, st2, symbols[(size_t)((cur2 << 8) | prv2)]);
                int cur3 = block[i3];
                encodeSymbol(n, st3, symbols[(size_t)((cur3 << 8) | prv3)]);
                prv0 = cur0; prv1 = cur1; prv2 = cur2; prv3 = cur3;
                i0--; i1--; i2--; i3--;
            }
            encodeSymbol(n, st0, symbols[(size_t)prv0]);
            encodeSymbol(n, st1, symbols[(size_t)prv1]);
            encodeSymbol(n, st2, symbols[(size_t)prv2]);
            encodeSymbol(n, st3, symbols[(size_t)prv3]);
        }
        n++;
        writeVarInt(bs, (uint32_t)((int64_t)buffer.size() - n));
        bs.writeBits((uint64_t)st0, 32);
        bs.writeBits((uint64_t)st1, 32);
        bs.writeBits((uint64_t)st2, 32);
        bs.writeBits((uint64_t)st3, 32);
        if ((int64_t)buffer.size() != n) bs.writeArray(&buffer[(size_t)n], 8 * (uint64_t)((int64_t)buffer.size() - n));
    }

    // :274-311
    void write(const uint8_t* block, size_t len) {
        if (len <= 32) { bs.writeArray(block, 8 * len); return; }
        size_t size = std::min<size_t>(2 * len, (size_t)chunkSize + ((size_t)chunkSize >> 3));
        size = std::max<size_t>(size, 65536);
        if (buffer.size() < size) buffer.assign(size, 0);
        size_t startChunk = 0;
        while (startChunk < len) {
            size_t endChunk = std::min(startChunk + (size_t)chunkSize, len);
            int alphabetSize = rebuildStatistics(block + startChunk, endChunk - startChunk, logRange);
            if (order == 1 || alphabetSize > 1) encodeChunk(block + startChunk, (int64_t)(endChunk - startChunk));
            startChunk = endChunk;
        }
    }
};

struct AnsDecSymbol { int64_t cumFreq, freq; };

struct AnsDecoder {
    BitReader& bs;
    unsigned order;
    unsigned logRange = 12;
    int chunkSize;
    std::vector<int64_t> freqs;       // dim*256
    std::vector<AnsDecSymbol> symbols;
    std::vector<uint8_t> f2s;
    std::vector<uint8_t> buffer;

    AnsDecoder(BitReader& b, unsigned ord) : bs(b), order(ord) {
        int chk = ANS0_CHUNK_SIZE;
        if (order == 1) chk = std::min(chk << 8, ANS_MAX_CHUNK_SIZE);
        int dim = (int)(255 * order + 1);
        freqs.assign((size_t)dim * 256, 0);
        symbols.resize((size_t)dim * 256);
        chunkSize = chk;
    }

    // :605-710
    int decodeHeader(int* alphabet) {
        logRange = (unsigned)(8 + bs.readBits(3));
        if (logRange < 8 || logRange > 16) throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: range");
        int res = 0;
        int dim = (int)(255 * order + 1);
        int64_t scale = (int64_t)1 << logRange;
        if (f2s.size() < (size_t)dim * (size_t)scale) f2s.assign((size_t)dim * (size_t)scale, 0);
        unsigned llr = 3;
        while (((unsigned)1 << llr) <= logRange) llr++;
        for (int k = 0; k < dim; k++) {
            int alphabetSize = decodeAlphabet(bs, alphabet, 256);
            if (alphabetSize == 0) continue;
            int64_t* f = &freqs[(size_t)k << 8];
            if (alphabetSize != 256) std::fill(f, f + 256, 0);
            int chkSize = alphabetSize < 64 ? 6 : 8;
            int64_t sum = 0;
            for (int i = 1; i < alphabetSize; i += chkSize) {
                unsigned logMax = (unsigned)bs.readBits(llr);
                if (((int64_t)1 << logMax) > scale) throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: incorrect frequency size");
                int endj = std::min(i + chkSize, alphabetSize);
                for (int j = i; j < endj; j++) {
                    int64_t freq = 1;
                    if (logMax > 0) {
                        freq = (int64_t)(1 + bs.readBits(logMax));
                        if (freq <= 0 || freq >= scale) throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: incorrect frequency");
                    }
                    f[alphabet[j]] = freq;
                    sum += freq;
                }
            }
            if (scale <= sum) throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: incorrect frequency");
            f[alphabet[0]] = scale - sum;
            sum = 0;
            AnsDecSymbol* symb = &symbols[(size_t)k << 8];
            uint8_t* freq2sym = &f2s[(size_t)k << logRange];
            for (int i = 0; i < 256; i++) {
                if (f[i] == 0) continue;
                for (int64_t j = f[i] - 1; j >= 0; j--) freq2sym[sum + j] = (uint8_t)i;
                symb[i].cumFreq = sum;                                           // decSymbol.reset :972-977
                symb[i].freq = std::min<int64_t>(f[i], ((int64_t)1 << logRange) - 1);
                sum += f[i];
            }
            res += alphabetSize;
        }
        return res;
    }

    // :846-858
    inline void decodeSymbol(int64_t& n, int64_t& st, const AnsDecSymbol& sym, int64_t mask) {
        st = sym.freq * (st >> logRange) + (st & mask) - sym.cumFreq;
        if (st < ANS_TOP) {
            st = (st << 16) | ((int64_t)buffer[(size_t)n] << 8) | (int64_t)buffer[(size_t)n + 1];
            n += 2;
        }
    }

    // :860-957
    bool decodeChunk(uint8_t* block, int64_t len) {
        uint32_t sz = readVarInt(bs);
        if (sz >= (uint32_t)ANS_MAX_CHUNK_SIZE) return false;
        int64_t st0 = (int64_t)bs.readBits(32), st1 = (int64_t)bs.readBits(32);
        int64_t st2 = (int64_t)bs.readBits(32), st3 = (int64_t)bs.readBits(32);
        if (len == 0) return true;
        size_t minBufSize = std::max<size_t>(2 * (size_t)len, 256);
        if (buffer.size() < minBufSize) buffer.assign(minBufSize, 0);
        if ((size_t)sz > buffer.size()) throw KnzError(ERR_PROCESS_BLOCK, "Invalid length"); // ReadArray panics
        bs.readArray(buffer.data(), 8 * (uint64_t)sz);
        if ((size_t)sz < buffer.size()) {
            size_t guardEnd = std::min((size_t)sz + 64, buffer.size());
            std::fill(buffer.begin() + sz, buffer.begin() + guardEnd, 0);
        }
        int64_t n = 0;
        int64_t mask = ((int64_t)1 << logRange) - 1;
        int64_t end4 = len & -4;
        auto chk = [&](int64_t nn) {
            if ((size_t)nn + 2 > buffer.size()) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
        };
        if (order == 0) {
            const uint8_t* freq2sym = f2s.data();
            const AnsDecSymbol* symb = symbols.data();
            for (int64_t i = 0; i < end4; i += 4) {
                chk(n + 6);
                uint8_t cur3 = freq2sym[st3 & mask]; block[i] = cur3; decodeSymbol(n, st3, symb[cur3], mask);
                uint8_t cur2 = freq2sym[st2 & mask]; block[i + 1] = cur2; decodeSymbol(n, st2, symb[cur2], mask);
                uint8_t cur1 = freq2sym[st1 & mask]; block[i + 2] = cur1; decodeSymbol(n, st1, symb[cur1], mask);
                uint8_t cur0 = freq2sym[st0 & mask]; block[i + 3] = cur0; decodeSymbol(n, st0, symb[cur0], mask);
            }
        } else {
            int64_t quarter = end4 >> 2;
            int64_t i0 = 0, i1 = quarter, i2 = 2 * quarter, i3 = 3 * quarter;
            int64_t prv0 = 0, prv1 = 0, prv2 = 0, prv3 = 0;
            while (i0 < quarter) {
                chk(n + 6);
                uint8_t cur3 = f2s[(size_t)((prv3 << logRange) + (st3 & mask))]; block[i3] = cur3;
                decodeSymbol(n, st3, symbols[(size_t)((prv3 << 8) + cur3)], mask);
                uint8_t cur2 = f2s[(size_t)((prv2 << logRange) + (st2 & mask))]; block[i2] = cur2;
                decodeSymbol(n, st2, symbols[(size_t)((prv2 << 8) + cur2)], mask);
                uint8_t cur1 = f2s[(size_t)((prv1 << logRange) + (st1 & mask))]; block[i1] = cur1;
                decodeSymbol(n, st1, symbols[(size_t)((prv1 << 8) + cur1)], mask);
                uint8_t cur0 = f2s[(size_t)((prv0 << logRange) + (st0 & mask))]; block[i0] = cur0;
                decodeSymbol(n, st0, symbols[(size_t)((prv0 << 8) + cur0)], mask);
                prv3 = cur3; prv2 = cur2; prv1 = cur1; prv0 = cur0;
                i0++; i1++; i2++; i3++;
            }
        }
        for (int64_t i = end4; i < len; i++) {
            if ((size_t)n >= buffer.size()) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
            block[i] = buffer[(size_t)n++];
        }
        return true;
    }

    // :714-757
    void read(uint8_t* block, size_t len) {
        if (len <= 32) { bs.readArray(block, 8 * len); return; }
        size_t startChunk = 0;
        int alphabet[256];
        while (startChunk < len) {
            size_t endChunk = std::min(startChunk + (size_t)chunkSize, len);
            int alphabetSize = decodeHeader(alphabet);
            if (alphabetSize == 0) throw KnzError(ERR_PROCESS_BLOCK, "ANS: empty alphabet");
            if (order == 0 && alphabetSize == 1) {
                memset(block + startChunk, alphabet[0], endChunk - startChunk);
            } else if (!decodeChunk(block + startChunk, (int64_t)(endChunk - startChunk))) {
                throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: incorrect chunk size");
            }
            startChunk = endChunk;
        }
    }
};

} // namespace knzo
