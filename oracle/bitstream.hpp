// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// CPU restatement of kanzi-go's MSB-first bit streams.
//   reference: v2/bitstream/DefaultOutputBitStream.go:78-272 (WriteBit/WriteBits/WriteArray/Close/Written)
//              v2/bitstream/DefaultInputBitStream.go:66-330  (ReadBit/ReadBits/ReadArray/pull)
// The reference buffers through an io.Writer; only the produced bit sequence is
// observable, so this restatement keeps a growing byte vector and a 64-bit
// accumulator. Go shift semantics (shift >= 64 yields 0) are reproduced by
// masking `value` to `count` bits before it is merged.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace knzo {

struct KnzError : std::runtime_error {
    int code;
    KnzError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// kanzi error codes, v2/Definitions.go:25-46
enum {
    ERR_MISSING_PARAM = 1, ERR_BLOCK_SIZE = 2, ERR_INVALID_CODEC = 3, ERR_CREATE_COMPRESSOR = 4,
    ERR_CREATE_DECOMPRESSOR = 5, ERR_OUTPUT_IS_DIR = 6, ERR_OVERWRITE_FILE = 7, ERR_CREATE_FILE = 8,
    ERR_CREATE_BITSTREAM = 9, ERR_OPEN_FILE = 10, ERR_READ_FILE = 11, ERR_WRITE_FILE = 12,
    ERR_PROCESS_BLOCK = 13, ERR_CREATE_CODEC = 14, ERR_INVALID_FILE = 15, ERR_STREAM_VERSION = 16,
    ERR_CREATE_STREAM = 17, ERR_INVALID_PARAM = 18, ERR_CRC_CHECK = 19, ERR_UNKNOWN = 127
};

class BitWriter {
public:
    std::vector<uint8_t> buf;   // completed bytes
    uint64_t cur = 0;           // pending bits, left-aligned
    unsigned fill = 0;          // number of pending bits in cur (0..63)

    void reserve(size_t n) { buf.reserve(n); }

    // DefaultOutputBitStream.go:66-76
    inline void writeBit(int bit) { writeBits((uint64_t)(bit & 1), 1); }

    // DefaultOutputBitStream.go:78-98 ; count in [0..64]
    inline void writeBits(uint64_t value, unsigned count) {
        if (count == 0) return;
        if (count > 64) throw KnzError(ERR_PROCESS_BLOCK, "Invalid bit count");
        if (count < 64) value &= ((uint64_t)1 << count) - 1;
        unsigned room = 64 - fill;
        if (count < room) {
            cur |= value << (room - count);
            fill += count;
        } else {
            unsigned rem = count - room;       // bits that do not fit
            uint64_t w = cur | (rem == 64 ? 0 : (value >> rem));
            push64(w);
            cur = (rem == 0) ? 0 : (value << (64 - rem));
            fill = rem;
        }
    }

    // DefaultOutputBitStream.go:101-199 : 'count' bits from 'bits', MSB first
    void writeArray(const uint8_t* bits, uint64_t count) {
        uint64_t nbytes = count >> 3;
        size_t i = 0;
        if ((fill & 7) == 0) {
            // byte aligned: drain accumulator then memcpy
            flushAccBytes();
            buf.insert(buf.end(), bits, bits + nbytes);
            i = nbytes;
        } else {
            for (; i + 8 <= nbytes; i += 8) {
                uint64_t v = be64(bits + i);
                uint64_t w = cur | (v >> fill);
                push64(w);
                cur = v << (64 - fill);
            }
            for (; i < nbytes; i++) writeBits(bits[i], 8);
        }
        unsigned r = (unsigned)(count & 7);
        if (r) writeBits((uint64_t)bits[i] >> (8 - r), r);
    }

    // Written(): exact number of bits so far. DefaultOutputBitStream.go:270-273
    uint64_t written() const { return ((uint64_t)buf.size() << 3) + fill; }

    // Close(): flush pending bits, zero-pad last byte. DefaultOutputBitStream.go:232-267
    // Returns the exact bit count (what Written() reports after Close()).
    uint64_t close() {
        uint64_t w = written();
        unsigned nb = (fill + 7) >> 3;
        for (unsigned k = 0; k < nb; k++) buf.push_back((uint8_t)(cur >> (56 - 8 * k)));
        cur = 0; fill = 0;
        return w;
    }

private:
    static inline uint64_t be64(const uint8_t* p) {
        uint64_t v; memcpy(&v, p, 8); return __builtin_bswap64(v);
    }
    inline void push64(uint64_t w) {
        size_t n = buf.size();
        buf.resize(n + 8);
        uint64_t s = __builtin_bswap64(w);
        memcpy(&buf[n], &s, 8);
    }
    inline void flushAccBytes() {
        unsigned nb = fill >> 3;
        for (unsigned k = 0; k < nb; k++) buf.push_back((uint8_t)(cur >> (56 - 8 * k)));
        cur = 0; fill = 0;
    }
};

// Reads from a byte buffer of 'nbytes' bytes. Reading past the last byte throws, as
// DefaultInputBitStream.pull() panics with "No more data to read in the bitstream"
// (DefaultInputBitStream.go:268-296, 216-250); callers turn that into ERR_PROCESS_BLOCK.
class BitReader {
public:
    const uint8_t* p;
    uint64_t nbits;   // total readable bits = 8*nbytes
    uint64_t pos = 0; // bit cursor

    BitReader(const uint8_t* data, uint64_t nbytes) : p(data), nbits(nbytes << 3) {}

    inline int readBit() { return (int)readBits(1); }

    // DefaultInputBitStream.go:78-96 ; count in [1..64]
    inline uint64_t readBits(unsigned count) {
        if (count == 0 || count > 64) throw KnzError(ERR_PROCESS_BLOCK, "Invalid bit count");
        if (pos + count > nbits) throw KnzError(ERR_PROCESS_BLOCK, "No more data to read in the bitstream");
        uint64_t byte = pos >> 3;
        unsigned off = (unsigned)(pos & 7);
        uint64_t res;
        uint64_t availBytes = (nbits >> 3) - byte;
        if (availBytes >= 9) {
            uint64_t hi = be64(p + byte);
            if (off + count <= 64) {
                res = (hi << off) >> (64 - count);
            } else {
                unsigned extra = off + count - 64;
                res = ((hi << off) >> (64 - count)) | ((uint64_t)p[byte + 8] >> (8 - extra));
            }
        } else {
            res = 0;
            for (unsigned k = 0; k < count; k++) {
                uint64_t bp = pos + k;
                res = (res << 1) | ((p[bp >> 3] >> (7 - (bp & 7))) & 1);
            }
        }
        pos += count;
        return res;
    }

    // DefaultInputBitStream.go:99-214
    void readArray(uint8_t* bits, uint64_t count) {
        if (count == 0) return;
        if (pos + count > nbits) throw KnzError(ERR_PROCESS_BLOCK, "No more data to read in the bitstream");
        uint64_t nbytes = count >> 3;
        if ((pos & 7) == 0) {
            memcpy(bits, p + (pos >> 3), nbytes);
            pos += nbytes << 3;
        } else {
            for (uint64_t i = 0; i < nbytes; i++) bits[i] = (uint8_t)readBits(8);
        }
        unsigned r = (unsigned)(count & 7);
        if (r) bits[nbytes] = (uint8_t)(readBits(r) << (8 - r));
    }

    uint64_t read() const { return pos; }

private:
    static inline uint64_t be64(const uint8_t* q) {
        uint64_t v; memcpy(&v, q, 8); return __builtin_bswap64(v);
    }
};

} // namespace knzo
