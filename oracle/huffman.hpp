// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// CPU restatement of kanzi-go's static Huffman codec (bitstream v6).
//   v2/entropy/HuffmanCodec.go:37-77   generateCanonicalCodes
//   :128-214 updateFrequencies ; :216-297 limitCodeLengths ; :300-326 computeCodeLengths
//   :328-385 computeInPlaceSizesPhase1/2 (Moffat-Katajainen)
//   :390-432 Write ; :435-511 encodeChunk
//   :620-657 readLengths ; :661-697 buildDecodingTable ; :758-969 decodeV6/decodeChunkV6
#pragma once
#include "entropy_utils.hpp"

namespace knzo {

static const int HUF_MAX_CHUNK_SIZE = 1 << 14;
static const int HUF_MAX_SYMBOL_SIZE = 12;

// HuffmanCodec.go:37-77. symbols (count entries) is re-ordered by (size, symbol).
static inline int hufGenerateCanonicalCodes(const uint8_t* sizes, uint16_t* codes, int* symbols, int count,
                                            int maxSymbolSize) {
    if (count == 0) return 0;
    if (count > 1) {
        static thread_local uint8_t buf[(HUF_MAX_SYMBOL_SIZE << 8) + 256];
        memset(buf, 0, sizeof(buf));
        for (int k = 0; k < count; k++) {
            int s = symbols[k];
            if (s > 255) throw KnzError(ERR_PROCESS_BLOCK, "Could not generate Huffman codes: invalid code length");
            if (sizes[s] > (uint8_t)maxSymbolSize)
                throw KnzError(ERR_PROCESS_BLOCK, "Could not generate Huffman codes: max code length exceeded");
            if (sizes[s] == 0) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); // (0-1)<<8 in Go panics
            buf[((int)(sizes[s] - 1) << 8) | s] = 1;
        }
        for (int i = 0, n = 0; n < count; i++) {
            symbols[n] = i & 0xFF;
            n += buf[i];
        }
    }
    uint16_t code = 0;
    uint8_t curLen = sizes[symbols[0]];
    for (int k = 0; k < count; k++) {
        int s = symbols[k];
        code = (uint16_t)(code << (sizes[s] - curLen));
        curLen = sizes[s];
        codes[s] = code;
        code++;
    }
    return count;
}

// :328-356
static inline void hufPhase1(int64_t* data, int n) {
    for (int s = 0, r = 0, t = 0; t < n - 1; t++) {
        int64_t sum = 0;
        for (int i = 0; i < 2; i++) {
            if (s >= n || (r < t && data[r] < data[s])) {
                sum += data[r];
                data[r] = t;
                r++;
                continue;
            }
            sum += data[s];
            if (s > t) data[s] = 0;
            s++;
        }
        data[t] = sum;
    }
}

// :359-385
static inline int hufPhase2(int64_t* data, int n) {
    if (n < 2) return 0;
    int levelTop = n - 2;
    int depth = 1;
    int i = n;
    int totalNodesAtLevel = 2;
    while (i > 0) {
        int k = levelTop;
        while (k > 0 && data[k - 1] >= levelTop) k--;
        int internalNodesAtLevel = levelTop - k;
        int leavesAtLevel = totalNodesAtLevel - internalNodesAtLevel;
        for (int j = 0; j < leavesAtLevel; j++) {
            i--;
            data[i] = depth;
        }
        totalNodesAtLevel = internalNodesAtLevel << 1;
        levelTop = k;
        depth++;
    }
    return depth - 1;
}

// :300-326. ranks holds (freq<<8|sym) keys on entry, symbols sorted by (freq,sym) on exit.
static inline int hufComputeCodeLengths(uint8_t* sizes, int64_t* ranks, int count) {
    int64_t freqs[256];
    std::sort(ranks, ranks + count);
    for (int i = 0; i < count; i++) {
        freqs[i] = ranks[i] >> 8;
        ranks[i] &= 0xFF;
        if (freqs[i] == 0) throw KnzError(ERR_PROCESS_BLOCK, "Could not generate Huffman codes: invalid code length 0");
    }
    hufPhase1(freqs, count);
    int maxCodeLen = hufPhase2(freqs, count);
    for (int i = 0; i < count; i++) sizes[ranks[i]] = (uint8_t)freqs[i];
    return maxCodeLen;
}

// :216-297
static inline int hufLimitCodeLengths(const int* symbols, int64_t* freqs, uint8_t* sizes, int64_t* ranks, int count) {
    int n = 0;
    int debt = 0;
    while (sizes[ranks[n]] >= HUF_MAX_SYMBOL_SIZE) {
        debt += (int)sizes[ranks[n]] - HUF_MAX_SYMBOL_SIZE;
        sizes[ranks[n]] = HUF_MAX_SYMBOL_SIZE;
        n++;
        if (n >= count) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
    }
    int64_t q[6][256];
    int qh[6] = {0}, qt[6] = {0};
    while (n < count) {
        uint8_t idx = (uint8_t)(HUF_MAX_SYMBOL_SIZE - 1 - sizes[ranks[n]]);
        if (idx > 5 || debt < (1 << idx)) break;
        q[idx][qt[idx]++] = ranks[n];
        n++;
    }
    int idx = 5;
    while (debt > 0 && idx >= 0) {
        if (qh[idx] == qt[idx] || debt < (1 << idx)) { idx--; continue; }
        int64_t r = q[idx][qh[idx]++];
        sizes[r]++;
        debt -= (1 << idx);
    }
    idx = 0;
    while (debt > 0 && idx < 6) {
        if (qh[idx] == qt[idx]) { idx++; continue; }
        int64_t r = q[idx][qh[idx]++];
        sizes[r]++;
        debt -= (1 << idx);
    }
    if (debt > 0) {
        int64_t f[256] = {0};
        int alpha[256] = {0};
        int64_t totalFreq = 0;
        for (int i = 0; i < count; i++) { f[i] = freqs[symbols[i]]; totalFreq += f[i]; }
        normalizeFrequencies(f, count, alpha, count, totalFreq, HUF_MAX_CHUNK_SIZE >> 3);
        for (int i = 0; i < count; i++) {
            freqs[symbols[i]] = f[i];
            ranks[i] = (f[i] << 8) | symbols[i];
        }
        return hufComputeCodeLengths(sizes, ranks, count);
    }
    return HUF_MAX_SYMBOL_SIZE;
}

struct HuffmanEncoder {
    BitWriter& bs;
    uint16_t codes[256];
    explicit HuffmanEncoder(BitWriter& b) : bs(b) { for (int i = 0; i < 256; i++) codes[i] = (uint16_t)i; }

    // :128-214
    int updateFrequencies(int64_t* freqs) {
        int count = 0;
        uint8_t sizes[256] = {0};
        int alphabet[256];
        for (int i = 0; i < 256; i++) {
            codes[i] = 0;
            if (freqs[i] > 0) alphabet[count++] = i;
        }
        encodeAlphabet(bs, alphabet, count);
        if (count == 0) return 0;
        if (count == 1) {
            codes[alphabet[0]] = 1 << 12;
            sizes[alphabet[0]] = 1;
        } else {
            int64_t ranks[256];
            for (int i = 0; i < count; i++) ranks[i] = (freqs[alphabet[i]] << 8) | alphabet[i];
            int maxCodeLen = hufComputeCodeLengths(sizes, ranks, count);
            if (maxCodeLen > HUF_MAX_SYMBOL_SIZE) maxCodeLen = hufLimitCodeLengths(alphabet, freqs, sizes, ranks, count);
            if (maxCodeLen > HUF_MAX_SYMBOL_SIZE) {
                for (int i = 0; i < count; i++) { codes[alphabet[i]] = (uint16_t)i; sizes[alphabet[i]] = 8; }
            } else {
                int syms[256];
                for (int i = 0; i < count; i++) syms[i] = (int)ranks[i];
                hufGenerateCanonicalCodes(sizes, codes, syms, count, HUF_MAX_SYMBOL_SIZE);
            }
        }
        uint8_t prevSize = 2;
        for (int k = 0; k < count; k++) {
            int s = alphabet[k];
            uint8_t curSize = sizes[s];
            codes[s] |= (uint16_t)((uint16_t)curSize << 12);
            expGolombEncodeByte(bs, (uint8_t)(curSize - prevSize));
            prevSize = curSize;
        }
        return count;
    }

    // :390-432
    void write(const uint8_t* block, size_t len) {
        if (len == 0) return;
        size_t startChunk = 0;
        while (startChunk < len) {
            size_t sizeChunk = std::min<size_t>(HUF_MAX_CHUNK_SIZE, len - startChunk);
            if (sizeChunk < 32) {
                bs.writeArray(block + startChunk, 8 * sizeChunk);
            } else {
                int64_t freqs[256] = {0};
                histogramO0(block + startChunk, sizeChunk, freqs);
                int count = updateFrequencies(freqs);
                if (count > 1) encodeChunk(block + startChunk, (int)sizeChunk);
            }
            startChunk += sizeChunk;
        }
    }

    // :435-511
    void encodeChunk(const uint8_t* block, int count) {
        uint32_t nbBits[4];
        int szFrag = count / 4;
        std::vector<uint8_t> frag[4];
        for (int j = 0; j < 4; j++) {
            const uint8_t* src = block + j * szFrag;
            BitWriter fw;
            fw.reserve((size_t)szFrag * 2 + 16);
            for (int i = 0; i < szFrag; i++) {
                uint16_t code = codes[src[i]];
                fw.writeBits((uint64_t)(code & 0x0FFF), code >> 12);
            }
            nbBits[j] = (uint32_t)fw.close();
            frag[j].swap(fw.buf);
            frag[j].push_back(0);
        }
        for (int j = 0; j < 4; j++) writeVarInt(bs, nbBits[j]);
        for (int j = 0; j < 4; j++) bs.writeArray(frag[j].data(), nbBits[j]);
        for (int i = 4 * szFrag; i < count; i++) bs.writeBits((uint64_t)block[i], 8);
    }
};

struct HuffmanDecoder {
    BitReader& bs;
    uint16_t codes[256];
    int alphabet[256];
    uint8_t sizes[256];
    uint16_t table[1 << HUF_MAX_SYMBOL_SIZE];
    explicit HuffmanDecoder(BitReader& b) : bs(b) {
        for (int i = 0; i < 256; i++) { sizes[i] = 8; codes[i] = (uint16_t)i; }
    }

    // :620-657
    int readLengths() {
        int count = decodeAlphabet(bs, alphabet, 256);
        if (count == 0) return 0;
        int8_t curSize = 2;
        for (int k = 0; k < count; k++) {
            int s = alphabet[k];
            codes[s] = 0;
            curSize = (int8_t)(curSize + (int8_t)expGolombDecodeByte(bs));
            if (curSize <= 0 || curSize > (int8_t)HUF_MAX_SYMBOL_SIZE)
                throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: incorrect size for Huffman symbol");
            sizes[s] = (uint8_t)curSize;
        }
        hufGenerateCanonicalCodes(sizes, codes, alphabet, count, HUF_MAX_SYMBOL_SIZE);
        return count;
    }

    // :661-697 ; alphabet is (size,symbol)-sorted after readLengths
    bool buildDecodingTable(int count) {
        for (int i = 0; i < (1 << HUF_MAX_SYMBOL_SIZE); i++) table[i] = 7;
        int length = 0;
        const int shift = HUF_MAX_SYMBOL_SIZE;
        for (int k = 0; k < count; k++) {
            int s = alphabet[k];
            if (sizes[s] > (uint8_t)length) length = sizes[s];
            uint16_t idx = (uint16_t)(codes[s] << (shift - length));
            uint16_t end = (uint16_t)(idx + (1 << (shift - length)));
            if ((int)end > (1 << HUF_MAX_SYMBOL_SIZE)) return false;
            uint16_t val = (uint16_t)(((uint16_t)s << 8) | sizes[s]);
            for (int j = idx; j < end; j++) table[j] = val;
        }
        return true;
    }

    // :758-805
    void read(uint8_t* block, size_t len) {
        if (len == 0) return;
        size_t startChunk = 0;
        while (startChunk < len) {
            size_t sizeChunk = std::min<size_t>(HUF_MAX_CHUNK_SIZE, len - startChunk);
            if (sizeChunk < 32) {
                bs.readArray(block + startChunk, 8 * sizeChunk);
            } else {
                int alphabetSize = readLengths();
                if (alphabetSize == 0) throw KnzError(ERR_PROCESS_BLOCK, "Huffman: empty alphabet"); // returns short count in Go
                if (alphabetSize == 1) {
                    memset(block + startChunk, alphabet[0], sizeChunk);
                } else {
                    if (!buildDecodingTable(alphabetSize))
                        throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: incorrect symbol size");
                    decodeChunk(block + startChunk, (int)sizeChunk);
                }
            }
            startChunk += sizeChunk;
        }
    }

    // :807-969. The reference decodes the four fragments in lock-step from zero-guarded
    // buffers; each fragment is an independent bit string so decoding them one after the
    // other yields the same bytes.
    void decodeChunk(uint8_t* block, int count) {
        uint32_t szBits[4];
        for (int j = 0; j < 4; j++) szBits[j] = readVarInt(bs);
        for (int j = 0; j < 4; j++)
            if ((int32_t)szBits[j] < 0) throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: incorrect stream size");
        int szFrag = count / 4;
        std::vector<uint8_t> buf;
        for (int j = 0; j < 4; j++) {
            size_t nb = ((size_t)szBits[j] + 7) >> 3;
            if (nb > (size_t)(2 * HUF_MAX_CHUNK_SIZE / 4))
                throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: fragment larger than decoder buffer");
            buf.assign(nb + 16, 0);
            bs.readArray(buf.data(), szBits[j]);
            uint8_t* dst = block + j * szFrag;
            uint64_t bitpos = 0;
            for (int n = 0; n < szFrag; n++) {
                size_t byte = (size_t)(bitpos >> 3);
                if (byte + 4 > buf.size()) throw KnzError(ERR_PROCESS_BLOCK, "Invalid bitstream: fragment overrun");
                uint32_t w = ((uint32_t)buf[byte] << 24) | ((uint32_t)buf[byte + 1] << 16) |
                             ((uint32_t)buf[byte + 2] << 8) | (uint32_t)buf[byte + 3];
                uint32_t idx = (w << (bitpos & 7)) >> (32 - HUF_MAX_SYMBOL_SIZE);
                uint16_t val = table[idx];
                bitpos += (uint8_t)val;
                dst[n] = (uint8_t)(val >> 8);
            }
        }
        for (int i = 4 * szFrag; i < count; i++) block[i] = (uint8_t)bs.readBits(8);
    }
};

} // namespace knzo
