// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// CPU restatement of kanzi-go's FPAQ codec (adaptive order-0 binary arithmetic coder) and
// the Null (pass-through) entropy codec.
//   v2/entropy/FPAQCodec.go:25-32 constants ; :100-117 encodeBit ; :123-171 Write ; :174-179 flush
//   :189-196 Dispose ; :308-334 decodeBitV2 ; :336-342 read ; :345-420 Read
//   v2/entropy/NullEntropyCodec.go:43-62 Write ; :91-110 Read
#pragma once
#include "entropy_utils.hpp"

namespace knzo {

static const int64_t FPAQ_PSCALE = 1 << 16;
static const size_t FPAQ_CHUNK = 4 * 1024 * 1024;
static const uint64_t FPAQ_TOP = 0x00FFFFFFFFFFFFFFull;
static const uint64_t FPAQ_MASK_0_56 = 0x00FFFFFFFFFFFFFFull;
static const uint64_t FPAQ_MASK_0_24 = 0x0000000000FFFFFFull;
static const uint64_t FPAQ_MASK_0_32 = 0x00000000FFFFFFFFull;

struct FpaqEncoder {
    BitWriter& bs;
    uint64_t low = 0, high = FPAQ_TOP;
    bool disposed = false;
    std::vector<uint8_t> buffer;
    size_t index = 0;
    int64_t probs[4][256];

    explicit FpaqEncoder(BitWriter& b) : bs(b) {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 256; j++) probs[i][j] = FPAQ_PSCALE >> 1;
    }

    // :174-179
    inline void flush() {
        uint32_t w = (uint32_t)(high >> 24);
        if (index + 4 > buffer.size()) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); // Go slice panic
        buffer[index] = (uint8_t)(w >> 24); buffer[index + 1] = (uint8_t)(w >> 16);
        buffer[index + 2] = (uint8_t)(w >> 8); buffer[index + 3] = (uint8_t)w;
        index += 4;
        low <<= 32;
        high = (high << 32) | FPAQ_MASK_0_32;
    }

    // :100-117
    inline void encodeBit(uint8_t bit, int64_t* p) {
        uint64_t split = (((high - low) >> 8) * (uint64_t)(*p)) >> 8;
        if (bit == 0) {
            low += split + 1;
            *p -= (*p >> 6);
        } else {
            high = low + split;
            *p -= ((*p - FPAQ_PSCALE + 64) >> 6);
        }
        if ((low ^ high) < ((uint64_t)1 << 24)) flush();
    }

    // :123-171
    void write(const uint8_t* block, size_t count) {
        if (count > ((size_t)1 << 30)) throw KnzError(ERR_PROCESS_BLOCK, "FPAQ codec: Invalid block size parameter");
        size_t startChunk = 0, end = count;
        while (startChunk < end) {
            size_t chunkSize = FPAQ_CHUNK;
            if (startChunk + FPAQ_CHUNK >= end) chunkSize = end - startChunk;
            if (buffer.size() < chunkSize + (chunkSize >> 3)) buffer.assign(chunkSize + (chunkSize >> 3), 0);
            index = 0;
            const uint8_t* buf = block + startChunk;
            int64_t* p = probs[0];
            for (size_t k = 0; k < chunkSize; k++) {
                uint8_t val = buf[k];
                int bits = (int)val + 256;
                encodeBit(val & 0x80, &p[1]);
                encodeBit(val & 0x40, &p[bits >> 7]);
                encodeBit(val & 0x20, &p[bits >> 6]);
                encodeBit(val & 0x10, &p[bits >> 5]);
                encodeBit(val & 0x08, &p[bits >> 4]);
                encodeBit(val & 0x04, &p[bits >> 3]);
                encodeBit(val & 0x02, &p[bits >> 2]);
                encodeBit(val & 0x01, &p[bits >> 1]);
                p = probs[val >> 6];
            }
            writeVarInt(bs, (uint32_t)index);
            bs.writeArray(buffer.data(), 8 * (uint64_t)index);
            startChunk += chunkSize;
            if (startChunk < end) bs.writeBits(low | FPAQ_MASK_0_24, 56);
        }
    }

    // :189-196
    void dispose() {
        if (disposed) return;
        disposed = true;
        bs.writeBits(low | FPAQ_MASK_0_24, 56);
    }
};

struct FpaqDecoder {
    BitReader& bs;
    uint64_t low = 0, high = FPAQ_TOP, current = 0;
    std::vector<uint8_t> buffer;
    size_t index = 0;
    int64_t probs[4][256];
    int ctx = 1; // Go: byte; kept wider, cast at use

    explicit FpaqDecoder(BitReader& b) : bs(b) {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 256; j++) probs[i][j] = FPAQ_PSCALE >> 1;
    }

    // :336-342
    inline void readWord() {
        low = (low << 32) & FPAQ_MASK_0_56;
        high = ((high << 32) | FPAQ_MASK_0_32) & FPAQ_MASK_0_56;
        if (index + 4 > buffer.size()) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
        uint64_t val = ((uint64_t)buffer[index] << 24) | ((uint64_t)buffer[index + 1] << 16) |
                       ((uint64_t)buffer[index + 2] << 8) | (uint64_t)buffer[index + 3];
        current = ((current << 32) | val) & FPAQ_MASK_0_56;
        index += 4;
    }

    // :308-334 ; ctx is a Go byte: ctx += ctx(+1) wraps mod 256
    inline void decodeBit(int64_t* p) {
        uint8_t c = (uint8_t)ctx;
        uint64_t split = ((((high - low) >> 8) * (uint64_t)p[c]) >> 8) + low;
        if (split >= current) {
            high = split;
            p[c] -= ((p[c] - FPAQ_PSCALE + 64) >> 6);
            ctx = (uint8_t)(c + c + 1);
        } else {
            low = split + 1; // -^split
            p[c] -= (p[c] >> 6);
            ctx = (uint8_t)(c + c);
        }
        if ((low ^ high) < ((uint64_t)1 << 24)) readWord();
    }

    // :345-420 (bsVersion >= 4 path)
    void read(uint8_t* block, size_t count) {
        if (count > ((size_t)1 << 30)) throw KnzError(ERR_PROCESS_BLOCK, "FPAQ codec: Invalid block size parameter");
        size_t startChunk = 0, end = count;
        while (startChunk < end) {
            int64_t szBytes = (int64_t)(int32_t)readVarInt(bs);
            if (szBytes < 0 || (size_t)szBytes >= 2 * count) throw KnzError(ERR_PROCESS_BLOCK, "FPAQ codec: Invalid chunk size");
            size_t bufSize = std::max<size_t>((size_t)(szBytes + (szBytes >> 2)), 1024);
            if (buffer.size() < bufSize) buffer.assign(bufSize, 0);
            current = bs.readBits(56);
            if ((size_t)szBytes < buffer.size()) {
                size_t guardEnd = std::min((size_t)szBytes + 8, buffer.size());
                std::fill(buffer.begin() + szBytes, buffer.begin() + guardEnd, 0);
            }
            bs.readArray(buffer.data(), 8 * (uint64_t)szBytes);
            index = 0;
            size_t chunkSize = std::min(FPAQ_CHUNK, end - startChunk);
            uint8_t* buf = block + startChunk;
            int64_t* p = probs[0];
            for (size_t i = 0; i < chunkSize; i++) {
                ctx = 1;
                decodeBit(p); decodeBit(p); decodeBit(p); decodeBit(p);
                decodeBit(p); decodeBit(p); decodeBit(p); decodeBit(p);
                buf[i] = (uint8_t)ctx;
                p = probs[(ctx & 0xFF) >> 6];
            }
            startChunk += chunkSize;
        }
    }
};

// NullEntropyCodec.go:43-62 / :91-110 (the 8 MiB piece size is not observable in the bits)
static inline void nullEntropyWrite(BitWriter& bs, const uint8_t* block, size_t count) { bs.writeArray(block, 8 * (uint64_t)count); }
static inline void nullEntropyRead(BitReader& bs, uint8_t* block, size_t count) { bs.readArray(block, 8 * (uint64_t)count); }

} // namespace knzo
