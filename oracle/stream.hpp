// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// CPU restatement of the kanzi-go block pipeline and .knz v6 stream framing.
//   v2/transform/Factory.go:26-53 ids, :58-95 New, :289-328 GetType
//   v2/transform/Sequence.go:64-125 Forward ; :131-186 Inverse ; :189-205 MaxEncodedLen
//   v2/entropy/EntropyCodecFactory.go:26-42 ids, :45-134 factories
//   v2/io/CompressedStream.go:429-519 writeHeader ; :729-977 encodingTask.encode ; :593-594 end marker
//   :1316-1522 readHeader ; :1763-2011 decodingTask.decode
//   v2/hash/XXHash32.go:51-102 ; v2/hash/XXHash64.go:51-120
#pragma once
#include "ans.hpp"
#include "fpaq.hpp"
#include "huffman.hpp"
#include "transforms.hpp"
#include <atomic>
#include <cmath>
#include <thread>

namespace knzo {

// transform ids (Factory.go:31-53)
enum : uint64_t { T_NONE = 0, T_BWT = 1, T_LZ = 3, T_ZRLT = 6, T_MTFT = 7, T_RANK = 8, T_TEXT = 10, T_SRT = 13, T_LZP = 14, T_LZX = 16, T_UTF = 17 };
// entropy ids (EntropyCodecFactory.go:26-42)
enum : uint32_t { E_NONE = 0, E_HUFFMAN = 1, E_FPAQ = 2, E_ANS0 = 5, E_ANS1 = 8 };

// XXHash32.go:51-102 (standard XXH32)
static inline uint32_t xxhash32(const uint8_t* data, size_t len, uint32_t seed) {
    const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
    auto rnd = [&](uint32_t acc, uint32_t val) { acc += val * P2; return ((acc << 13) | (acc >> 19)) * P1; };
    size_t end = len, n = 0;
    uint32_t h32;
    if (end >= 16) {
        size_t end16 = end - 16;
        uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        while (n <= end16) {
            v1 = rnd(v1, le32(data + n)); v2 = rnd(v2, le32(data + n + 4));
            v3 = rnd(v3, le32(data + n + 8)); v4 = rnd(v4, le32(data + n + 12));
            n += 16;
        }
        h32 = ((v1 << 1) | (v1 >> 31)) + ((v2 << 7) | (v2 >> 25)) + ((v3 << 12) | (v3 >> 20)) + ((v4 << 18) | (v4 >> 14));
    } else h32 = seed + P5;
    h32 += (uint32_t)end;
    while (n + 4 <= end) { h32 += le32(data + n) * P3; h32 = ((h32 << 17) | (h32 >> 15)) * P4; n += 4; }
    while (n < end) { h32 += (uint32_t)data[n] * P5; h32 = ((h32 << 11) | (h32 >> 21)) * P1; n++; }
    h32 ^= h32 >> 15; h32 *= P2; h32 ^= h32 >> 13; h32 *= P3;
    return h32 ^ (h32 >> 16);
}

// XXHash64.go:51-120. NOTE: the merge of v1..v4 uses 32-bit style shift pairs on 64-bit words
// ((v1<<1)|(v1>>31) ...), i.e. it is NOT standard XXH64; restated literally.
static inline uint64_t xxhash64(const uint8_t* data, size_t len, uint64_t seed) {
    const uint64_t P1 = 0x9E3779B185EBCA87ull, P2 = 0xC2B2AE3D27D4EB4Full, P3 = 0x165667B19E3779F9ull,
                   P4 = 0x85EBCA77C2B2AE63ull, P5 = 0x27D4EB2F165667C5ull;
    auto rnd = [&](uint64_t acc, uint64_t val) { acc += val * P2; return ((acc << 31) | (acc >> 33)) * P1; };
    auto mrg = [&](uint64_t acc, uint64_t val) { acc ^= rnd(0, val); return acc * P1 + P4; };
    size_t end = len, n = 0;
    uint64_t h64;
    if (end >= 32) {
        size_t end32 = end - 32;
        uint64_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        while (n <= end32) {
            v1 = rnd(v1, le64(data + n)); v2 = rnd(v2, le64(data + n + 8));
            v3 = rnd(v3, le64(data + n + 16)); v4 = rnd(v4, le64(data + n + 24));
            n += 32;
        }
        h64 = ((v1 << 1) | (v1 >> 31)) + ((v2 << 7) | (v2 >> 25)) + ((v3 << 12) | (v3 >> 20)) + ((v4 << 18) | (v4 >> 14));
        h64 = mrg(h64, v1); h64 = mrg(h64, v2); h64 = mrg(h64, v3); h64 = mrg(h64, v4);
    } else h64 = seed + P5;
    h64 += (uint64_t)end;
    while (n + 8 <= end) { h64 ^= rnd(0, le64(data + n)); h64 = ((h64 << 27) | (h64 >> 37)) * P1 + P4; n += 8; }
    while (n + 4 <= end) { h64 ^= (uint64_t)le32(data + n) * P1; h64 = ((h64 << 23) | (h64 >> 41)) * P2 + P3; n += 4; }
    while (n < end) { h64 += (uint64_t)data[n] * P5; h64 = ((h64 << 11) | (h64 >> 53)) * P1; n++; }
    h64 ^= h64 >> 33; h64 *= P2; h64 ^= h64 >> 29; h64 *= P3;
    return h64 ^ (h64 >> 32);
}

// ---- single transform dispatch (Factory.go:97-185 newToken) ---------------------------------
static inline bool transformSupported(uint64_t t) {
    return t == T_NONE || t == T_BWT || t == T_LZ || t == T_LZX || t == T_ZRLT || t == T_MTFT || t == T_RANK || t == T_SRT || t == T_LZP || t == T_UTF || t == T_TEXT;
}
static inline size_t transformMaxEncodedLen(uint64_t t, size_t n) {
    switch (t) {
        case T_BWT: case T_RANK: case T_MTFT: return n + BWT_MAX_HEADER_SIZE;
        case T_LZ: case T_LZX: case T_LZP: return lzMaxEncodedLen(n);
        case T_SRT: return n + SRT_MAX_HEADER_SIZE;
        case T_UTF: return n + 8192;
        default: return n;
    }
}
// throws SkipTransform if the transform declines
static inline size_t transformForward1(uint64_t t, const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    switch (t) {
        case T_NONE:
            if (cap < n) throw SkipTransform("Output buffer is too small");
            if (n) memcpy(dst, src, n);
            return n;
        case T_BWT: return bwtBlockForward(src, n, dst, cap);
        case T_LZ: return lzForward(src, n, dst, cap, false);
        case T_LZX: return lzForward(src, n, dst, cap, true);
        case T_ZRLT: return zrltForward(src, n, dst, cap);
        case T_MTFT: return SBRT(1).forward(src, n, dst, cap);
        case T_RANK: return SBRT(2).forward(src, n, dst, cap);
        case T_SRT: return srtForward(src, n, dst, cap);
        case T_LZP: return lzpForward(src, n, dst, cap);
        case T_UTF: return utfForward(src, n, dst, cap);
        case T_TEXT: return textForward(src, n, dst, cap);
        default: throw KnzError(ERR_CREATE_CODEC, "Unknown transform type");
    }
}
static inline size_t transformInverse1(uint64_t t, const uint8_t* src, size_t n, uint8_t* dst, size_t cap) {
    switch (t) {
        case T_NONE:
            if (n > cap) throw KnzError(ERR_PROCESS_BLOCK, "Destination buffer too small");
            if (n) memcpy(dst, src, n);
            return n;
        case T_BWT: return bwtBlockInverse(src, n, dst, cap);
        case T_LZ: case T_LZX: return lzInverse(src, n, dst, cap);
        case T_ZRLT: return zrltInverse(src, n, dst, cap);
        case T_MTFT: return SBRT(1).inverse(src, n, dst, cap);
        case T_RANK: return SBRT(2).inverse(src, n, dst, cap);
        case T_SRT: return srtInverse(src, n, dst, cap);
        case T_LZP: return lzpInverse(src, n, dst, cap);
        case T_UTF: return utfInverse(src, n, dst, cap);
        case T_TEXT: return textInverse(src, n, dst, cap);
        default: throw KnzError(ERR_INVALID_CODEC, "Unknown transform type");
    }
}

// ---- ByteTransformSequence -------------------------------------------------------------------
struct Sequence {
    std::vector<uint64_t> transforms;
    uint8_t skipFlags = 0;

    // Factory.go:58-95
    explicit Sequence(uint64_t functionType) {
        int nbtr = 0;
        for (int s = 42; s >= 0; s -= 6) if (((functionType >> s) & 63) != T_NONE) nbtr++;
        if (nbtr == 0) nbtr = 1;
        for (int i = 0; i < nbtr; i++) {
            uint64_t t = (functionType >> (42 - 6 * i)) & 63;
            if (!transformSupported(t)) throw KnzError(ERR_CREATE_CODEC, "Unknown transform type (oracle covers the hot-path transforms only)");
            transforms.push_back(t);
        }
    }
    int len() const { return (int)transforms.size(); }

    // Sequence.go:189-205
    size_t maxEncodedLen(size_t srcLen) const {
        size_t req = srcLen;
        for (uint64_t t : transforms) req = std::max(req, transformMaxEncodedLen(t, req));
        return req;
    }

    // Sequence.go:64-125. dst must hold maxEncodedLen(n). Returns output length.
    size_t forward(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {
        skipFlags = 0xFF;
        if (n == 0 || dstCap == 0) return 0;
        size_t requiredSize = maxEncodedLen(n);
        if (dstCap < requiredSize) throw KnzError(ERR_PROCESS_BLOCK, "Output buffer is too small");
        std::vector<uint8_t> tmp; // second ping-pong buffer (the reference re-uses src's backing array)
        const uint8_t* in = src;
        uint8_t* out = dst;
        size_t length = n;
        int swaps = 0;
        bool inIsDst = false;
        for (int i = 0; i < len(); i++) {
            size_t saved = length;
            size_t outCap = (out == dst) ? dstCap : tmp.size();
            try {
                length = transformForward1(transforms[i], in, length, out, outCap);
            } catch (const SkipTransform&) {
                length = saved;
                continue;
            }
            skipFlags &= (uint8_t)~(1u << (7 - i));
            // swap
            if (out == dst) {
                in = dst; inIsDst = true;
                if (tmp.size() < requiredSize) tmp.resize(requiredSize);
                out = tmp.data();
            } else {
                in = tmp.data(); inIsDst = false;
                out = dst;
            }
            swaps++;
        }
        if ((swaps & 1) == 0) {
            if (dstCap < length) skipFlags = 0xFF;
            else if (!inIsDst && length) memmove(dst, in, length);
        }
        return length;
    }

    // Sequence.go:131-186
    size_t inverse(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {
        if (n == 0 || dstCap == 0) return 0;
        if (skipFlags == 0xFF) {
            if (n > dstCap) throw KnzError(ERR_PROCESS_BLOCK, "Inverse transform sequence failed");
            memcpy(dst, src, n);
            return n;
        }
        std::vector<uint8_t> tmp;
        const uint8_t* in = src;
        uint8_t* out = dst;
        size_t length = n;
        int swaps = 0;
        bool inIsDst = false;
        for (int i = len() - 1; i >= 0; i--) {
            if (skipFlags & (1u << (7 - i))) continue;
            size_t outCap = dstCap;
            if (out != dst) { if (tmp.size() < dstCap) tmp.resize(dstCap); out = tmp.data(); }
            length = transformInverse1(transforms[i], in, length, out, outCap);
            if (out == dst) { in = dst; inIsDst = true; if (tmp.size() < dstCap) tmp.resize(dstCap); out = tmp.data(); }
            else { in = tmp.data(); inIsDst = false; out = dst; }
            swaps++;
        }
        if ((swaps & 1) == 0) {
            if (dstCap < length) throw KnzError(ERR_PROCESS_BLOCK, "Inverse transform sequence failed");
            if (!inIsDst && length) memmove(dst, in, length);
        }
        return length;
    }
};

// ---- entropy dispatch -------------------------------------------------------------------------
static inline void entropyEncode(BitWriter& obs, uint32_t type, const uint8_t* block, size_t n) {
    switch (type) {
        case E_NONE: nullEntropyWrite(obs, block, n); break;
        case E_HUFFMAN: { HuffmanEncoder e(obs); e.write(block, n); break; }
        case E_ANS0: { AnsEncoder e(obs, 0); e.write(block, n); break; }
        case E_ANS1: { AnsEncoder e(obs, 1); e.write(block, n); break; }
        case E_FPAQ: { FpaqEncoder e(obs); e.write(block, n); e.dispose(); break; }
        default: throw KnzError(ERR_CREATE_CODEC, "Unsupported entropy codec type (oracle covers the hot-path codecs only)");
    }
}
static inline void entropyDecode(BitReader& ibs, uint32_t type, uint8_t* block, size_t n) {
    switch (type) {
        case E_NONE: nullEntropyRead(ibs, block, n); break;
        case E_HUFFMAN: { HuffmanDecoder d(ibs); d.read(block, n); break; }
        case E_ANS0: { AnsDecoder d(ibs, 0); d.read(block, n); break; }
        case E_ANS1: { AnsDecoder d(ibs, 1); d.read(block, n); break; }
        case E_FPAQ: { FpaqDecoder d(ibs); d.read(block, n); break; }
        default: throw KnzError(ERR_INVALID_CODEC, "Unsupported entropy codec type (oracle covers the hot-path codecs only)");
    }
}

// ---- `-s` / ctx["skipBlocks"] (CompressedStream.go:778-800): blocks that look incompressible become copy blocks -------------
// internal/Magic.go:83-126 GetMagicType, :130-170 IsDataCompressed
static inline uint32_t getMagicType(const uint8_t* src, size_t n) {
    if (n < 4) return 0;
    const uint32_t key = ((uint32_t)src[0] << 24) | ((uint32_t)src[1] << 16) | ((uint32_t)src[2] << 8) | src[3];
    if ((key & ~0x0Fu) == 0xFFD8FFE0u) return key;                       // JPG
    if ((key >> 8) == 0x425A68u || (key >> 8) == 0x494433u) return key >> 8;   // BZIP2, MP3 ID3
    static const uint32_t keys32[18] = {0x47494638u, 0x25504446u, 0x504B0304u, 0x377ABCAFu, 0x89504E47u, 0x7F454C46u, 0xFEEDFACEu, 0xCEFAEDFEu,
                                        0xFEEDFACFu, 0xCFFAEDFEu, 0x28B52FFDu, 0x81CFB2CEu, 0x4D534346u, 0x52494646u, 0x664C6143u, 0xFD377A58u,
                                        0x4B414E5Au, 0x52617221u};
    for (uint32_t k : keys32) if (key == k) return key;
    const uint32_t key16 = key >> 16;
    if (key16 == 0x1F8Bu || key16 == 0x424Du || key16 == 0x4D5Au) return key16;      // GZIP, BMP, WIN
    if (key16 == 0x5034u || key16 == 0x5035u || key16 == 0x5036u) {                  // PBM, PGM, PPM (binary)
        const uint32_t sub = (key >> 8) & 0xFF;
        if (sub == 0x07 || sub == 0x0A || sub == 0x0D || sub == 0x20) return key16;
    }
    return 0;
}
static inline bool isDataMultimedia(uint32_t magic) {   // Magic.go:174-200
    switch (magic) {
        case 0xFFD8FFE0u: case 0x47494638u: case 0x89504E47u: case 0x52494646u: case 0x664C6143u: case 0x494433u: case 0x424Du:
        case 0x5034u: case 0x5035u: case 0x5036u: return true;     // JPG GIF PNG RIFF FLAC MP3 BMP PBM PGM PPM
        default: return false;
    }
}
static inline bool isDataExecutable(uint32_t magic) {   // Magic.go:204-222
    switch (magic) {
        case 0x7F454C46u: case 0x4D5Au: case 0xFEEDFACEu: case 0xCEFAEDFEu: case 0xFEEDFACFu: case 0xCFFAEDFEu: return true;   // ELF WIN MAC x4
        default: return false;
    }
}
static inline bool isDataCompressed(uint32_t magic) {
    switch (magic) {
        case 0xFFD8FFE0u: case 0x47494638u: case 0x89504E47u: case 0x377ABCAFu: case 0x28B52FFDu: case 0x81CFB2CEu: case 0x4D534346u:
        case 0x504B0304u: case 0x1F8Bu: case 0x425A68u: case 0x664C6143u: case 0x494433u: case 0xFD377A58u: case 0x4B414E5Au: case 0x52617221u:
            return true;                 // JPG (exact value only, as in the reference's switch), GIF, PNG, LZMA, ZSTD, BROTLI, CAB, ZIP, GZIP, BZIP2, FLAC, MP3, XZ, KNZ, RAR
        default: return false;
    }
}
// internal/Global.go:174-191 Log2ScaledBy1024 ; the table is 4096*log2(x) rounded, x = 0..256 (:59-88)
static inline uint32_t log2ScaledBy1024(uint32_t x) {
    static uint32_t table[257];
    static bool init = false;
    if (!init) { table[0] = 0; for (int i = 1; i <= 256; i++) table[i] = (uint32_t)std::floor(4096.0 * std::log2((double)i) + 0.5); init = true; }
    if (x < 256) return (table[x] + 2) >> 2;
    const uint32_t lg = log2NoCheck(x);
    if ((x & (x - 1)) == 0) return lg << 10;
    return ((lg - 7) * 1024) + ((table[x >> (lg - 7)] + 2) >> 2);
}
// internal/Global.go:196-216
static inline int computeFirstOrderEntropy1024(size_t blockLen, const int* histo) {
    if (blockLen == 0) return 0;
    uint64_t sum = 0;
    const uint32_t logLength1024 = log2ScaledBy1024((uint32_t)blockLen);
    for (int i = 0; i < 256; i++) {
        if (histo[i] == 0) continue;
        const uint32_t log1024 = log2ScaledBy1024((uint32_t)histo[i]);
        sum += ((uint64_t)histo[i] * (uint64_t)(logLength1024 - log1024)) >> 3;
    }
    return (int)(sum / (uint64_t)blockLen);
}
static const int INCOMPRESSIBLE_THRESHOLD = 973;   // entropy/EntropyUtils.go:26
enum : int { KNZO_FLAG_SKIP_BLOCKS = 1 };

// ---- one block: CompressedStream.go:729-914 (up to obs.Close()) -------------------------------
struct BlockResult {
    std::vector<uint8_t> bits; // block-local stream, zero padded to a byte
    uint64_t written = 0;      // exact bit count
    size_t postLen = 0;
    uint8_t skipFlags = 0, mode = 0;
    uint64_t checksum = 0;
};

static const uint32_t KNZ_SEED = 0x4B414E5A;

static inline void encodeBlock(const uint8_t* data, size_t blockLength, uint64_t transformType, uint32_t entropyType,
                               int checksumBits, BlockResult& res, int flags = 0) {
    uint8_t mode = 0;
    uint64_t checksum = 0;
    if (checksumBits == 32) checksum = xxhash32(data, blockLength, KNZ_SEED);
    else if (checksumBits == 64) checksum = xxhash64(data, blockLength, KNZ_SEED);
    if (blockLength <= 15) { // _SMALL_BLOCK_SIZE :773-776
        transformType = T_NONE;
        entropyType = E_NONE;
        mode |= 0x80;
    } else if (flags & KNZO_FLAG_SKIP_BLOCKS) { // :778-800
        bool skip = false;
        if (blockLength >= 8) skip = isDataCompressed(getMagicType(data, blockLength));
        if (!skip) {
            int histo[256] = {0};
            for (size_t i = 0; i < blockLength; i++) histo[data[i]]++;
            skip = computeFirstOrderEntropy1024(blockLength, histo) >= INCOMPRESSIBLE_THRESHOLD;
        }
        if (skip) { transformType = T_NONE; entropyType = E_NONE; mode |= 0x80; }
    }
    {   // ctx["dataType"] from the block's magic number (:811-819); read by the UTF codec only on this path
        const uint32_t magic = getMagicType(data, blockLength);
        tlsDataType = DT_UNDEFINED;
        if (isDataCompressed(magic)) tlsDataType = DT_BIN;
        else if (isDataMultimedia(magic)) tlsDataType = DT_MULTIMEDIA;
        else if (isDataExecutable(magic)) tlsDataType = DT_EXE;
    }
    Sequence t(transformType);
    size_t requiredSize = t.maxEncodedLen(blockLength);
    std::vector<uint8_t> buffer(std::max<size_t>(requiredSize, 1));
    size_t postTransformLength = t.forward(data, blockLength, buffer.data(), buffer.size());
    unsigned dataSize = 1;
    if (postTransformLength >= 256) {
        dataSize = (log2NoCheck((uint32_t)postTransformLength) >> 3) + 1;
        if (dataSize > 4) throw KnzError(ERR_WRITE_FILE, "Invalid block data length");
    }
    mode |= (uint8_t)(((dataSize - 1) & 0x03) << 5);
    BitWriter obs;
    obs.reserve(postTransformLength + (postTransformLength >> 3) + 64);
    uint8_t skipFlags = t.skipFlags;
    if ((mode & 0x80) != 0 || t.len() <= 4) {
        mode |= (uint8_t)(skipFlags >> 4);
        obs.writeBits(mode, 8);
    } else {
        mode |= 0x10;
        obs.writeBits(mode, 8);
        obs.writeBits(skipFlags, 8);
    }
    obs.writeBits((uint64_t)postTransformLength, 8 * dataSize);
    if (checksumBits == 32) obs.writeBits(checksum, 32);
    else if (checksumBits == 64) obs.writeBits(checksum, 64);
    entropyEncode(obs, entropyType, buffer.data(), postTransformLength);
    res.written = obs.close();
    res.bits.swap(obs.buf);
    res.postLen = postTransformLength;
    res.skipFlags = skipFlags;
    res.mode = mode;
    res.checksum = checksum;
}

// CompressedStream.go:1875-2011 (after the payload has been read). blockSize = stream block size.
static inline size_t decodeBlock(const uint8_t* payload, size_t payloadBytes, uint64_t transformType, uint32_t entropyType,
                                 int checksumBits, size_t blockSize, uint8_t* out, size_t outCap) {
    BitReader ibs(payload, payloadBytes);
    uint8_t mode = (uint8_t)ibs.readBits(8);
    uint8_t skipFlags = 0;
    if (mode & 0x80) { transformType = T_NONE; entropyType = E_NONE; }
    else if (mode & 0x10) skipFlags = (uint8_t)ibs.readBits(8);
    else skipFlags = (uint8_t)((mode << 4) | 0x0F);
    unsigned dataSize = 1 + ((mode >> 5) & 0x03);
    unsigned length = dataSize << 3;
    uint64_t mask = ((uint64_t)1 << length) - 1;
    size_t preTransformLength = (size_t)(ibs.readBits(length) & mask);
    size_t maxTransformLength = std::min<size_t>(std::max<size_t>(blockSize + blockSize / 2, 2048), (size_t)1 << 30);
    if (preTransformLength == 0 || preTransformLength > maxTransformLength) throw KnzError(ERR_BLOCK_SIZE, "Invalid compressed block size");
    uint64_t checksum1 = 0;
    if (checksumBits == 32) checksum1 = ibs.readBits(32);
    else if (checksumBits == 64) checksum1 = ibs.readBits(64);
    std::vector<uint8_t> buffer(std::max(blockSize, preTransformLength + 512));
    entropyDecode(ibs, entropyType, buffer.data(), preTransformLength);
    Sequence t(transformType);
    t.skipFlags = skipFlags;
    size_t decoded = t.inverse(buffer.data(), preTransformLength, out, outCap);
    if (checksumBits == 32) {
        if (xxhash32(out, decoded, KNZ_SEED) != (uint32_t)checksum1) throw KnzError(ERR_CRC_CHECK, "Corrupted bitstream: checksum mismatch");
    } else if (checksumBits == 64) {
        if (xxhash64(out, decoded, KNZ_SEED) != checksum1) throw KnzError(ERR_CRC_CHECK, "Corrupted bitstream: checksum mismatch");
    }
    return decoded;
}

// ---- stream header: CompressedStream.go:429-519 -----------------------------------------------
static inline uint32_t headerChecksum(int ckSize, uint32_t entropyType, uint64_t transformType, int64_t blockSize,
                                      unsigned szMask, int64_t inputSize) {
    const uint32_t HASH = 0x1E35A7BDu;
    uint32_t seed = 0x01030507u * 6u;
    uint32_t cksum = HASH * seed;
    cksum ^= HASH * (uint32_t)(~(int64_t)ckSize);
    cksum ^= HASH * (uint32_t)(~entropyType);
    cksum ^= HASH * (uint32_t)((~transformType) >> 32);
    cksum ^= HASH * (uint32_t)(~transformType);
    cksum ^= HASH * (uint32_t)(~blockSize);
    if (szMask > 0) {
        cksum ^= HASH * (uint32_t)((~inputSize) >> 32);   // arithmetic shift on int64, as in Go
        cksum ^= HASH * (uint32_t)(~inputSize);
    }
    return ((cksum >> 23) ^ (cksum >> 3)) & 0xFFFFFF;
}

static inline void writeHeader(BitWriter& obs, int checksumBits, uint32_t entropyType, uint64_t transformType,
                               int64_t blockSize, int64_t inputSize) {
    int ckSize = checksumBits == 32 ? 1 : (checksumBits == 64 ? 2 : 0);
    obs.writeBits(0x4B414E5A, 32);
    obs.writeBits(6, 4);
    obs.writeBits((uint64_t)ckSize, 2);
    obs.writeBits(entropyType, 5);
    obs.writeBits(transformType, 48);
    obs.writeBits((uint64_t)(blockSize >> 4), 28);
    unsigned szMask;
    if (inputSize == 0) szMask = 0;
    else if (inputSize >= ((int64_t)1 << 48)) szMask = 0;
    else if (inputSize >= ((int64_t)1 << 32)) szMask = 3;
    else if (inputSize >= ((int64_t)1 << 16)) szMask = 2;
    else szMask = 1;
    obs.writeBits(szMask, 2);
    if (szMask > 0) obs.writeBits((uint64_t)inputSize, 16 * szMask);
    obs.writeBits(0, 15);
    obs.writeBits(headerChecksum(ckSize, entropyType, transformType, blockSize, szMask, inputSize), 24);
}

struct StreamHeader { int checksumBits; uint32_t entropyType; uint64_t transformType; int64_t blockSize; int64_t outputSize; unsigned szMask; };

static inline StreamHeader readHeader(BitReader& ibs) {
    StreamHeader h{};
    if (ibs.readBits(32) != 0x4B414E5A) throw KnzError(ERR_INVALID_FILE, "Invalid stream type");
    unsigned bsVersion = (unsigned)ibs.readBits(4);
    if (bsVersion != 6) throw KnzError(ERR_STREAM_VERSION, "oracle reads bitstream version 6 only");
    uint64_t ckSize = ibs.readBits(2);
    if (ckSize == 3) throw KnzError(ERR_INVALID_CODEC, "Invalid bitstream, incorrect checksum size");
    h.checksumBits = ckSize == 1 ? 32 : (ckSize == 2 ? 64 : 0);
    h.entropyType = (uint32_t)ibs.readBits(5);
    h.transformType = ibs.readBits(48);
    h.blockSize = (int64_t)ibs.readBits(28) << 4;
    if (h.blockSize < 1024 || h.blockSize > 1024 * 1024 * 1024) throw KnzError(ERR_BLOCK_SIZE, "Invalid bitstream, incorrect block size");
    h.szMask = (unsigned)ibs.readBits(2);
    if (h.szMask != 0) h.outputSize = (int64_t)ibs.readBits(16 * h.szMask);
    ibs.readBits(15);
    uint32_t cksum1 = (uint32_t)ibs.readBits(24);
    uint32_t cksum2 = headerChecksum((int)ckSize, h.entropyType, h.transformType, h.blockSize, h.szMask, h.outputSize);
    if (cksum1 != cksum2) throw KnzError(ERR_CRC_CHECK, "Invalid bitstream: checksum mismatch");
    return h;
}

// ---- whole stream: Writer.Write/Close (:524-620) and Reader (:1556-1760), single pass -----------
// 'jobs' worker threads each take whole blocks (the reference runs one goroutine per block,
// :658-698); emission is in block order (:934-976).
static inline void appendBlock(BitWriter& obs, const BlockResult& r) {
    uint64_t written = r.written;
    unsigned lw = 3;
    if (written >= 8) lw = log2NoCheck((uint32_t)(written >> 3)) + 4;
    obs.writeBits(lw - 3, 5);
    obs.writeBits(written, lw);
    obs.writeArray(r.bits.data(), written); // the 1<<30-bit piece size (:961-976) is not observable
}

static inline void compressStream(const uint8_t* src, size_t n, uint64_t transformType, uint32_t entropyType,
                                  size_t blockSize, int checksumBits, int jobs, int64_t headerInputSize,
                                  std::vector<uint8_t>& out, int flags = 0) {
    if (blockSize < 1024 || blockSize > ((size_t)1 << 30) || (blockSize & 15)) throw KnzError(ERR_BLOCK_SIZE, "Invalid block size");
    size_t nblocks = (n + blockSize - 1) / blockSize;
    std::vector<BlockResult> results(nblocks);
    std::atomic<size_t> next(0);
    std::atomic<int> errCode(0);
    std::string errMsg;
    auto worker = [&]() {
        tlsBlockSize = (uint32_t)blockSize; tlsEntropyType = entropyType;       // ctx["blockSize"], ctx["entropy"] (:218-220)
        for (;;) {
            size_t b = next.fetch_add(1);
            if (b >= nblocks || errCode.load()) return;
            size_t off = b * blockSize, len = std::min(blockSize, n - off);
            try { encodeBlock(src + off, len, transformType, entropyType, checksumBits, results[b], flags); }
            catch (const KnzError& e) { int z = 0; if (errCode.compare_exchange_strong(z, e.code)) errMsg = e.what(); return; }
        }
    };
    if (jobs <= 1) worker();
    else { std::vector<std::thread> th; for (int j = 0; j < jobs; j++) th.emplace_back(worker); for (auto& t : th) t.join(); }
    if (errCode.load()) throw KnzError(errCode.load(), errMsg);
    BitWriter obs;
    obs.reserve(n / 2 + 64);
    writeHeader(obs, checksumBits, entropyType, transformType, (int64_t)blockSize, headerInputSize);
    for (size_t b = 0; b < nblocks; b++) { appendBlock(obs, results[b]); std::vector<uint8_t>().swap(results[b].bits); }
    obs.writeBits(0, 5); // end marker :593-594
    obs.writeBits(0, 3);
    obs.close();
    out.swap(obs.buf);
}

static inline void decompressStream(const uint8_t* src, size_t n, int jobs, std::vector<uint8_t>& out) {
    BitReader ibs(src, n);
    StreamHeader h = readHeader(ibs);
    struct Payload { std::vector<uint8_t> data; };
    std::vector<Payload> payloads;
    for (;;) { // :1816-1852
        unsigned lr = (unsigned)ibs.readBits(5) + 3;
        uint64_t read = ibs.readBits(lr);
        if (read == 0) break;
        if (read > ((uint64_t)1 << 34)) throw KnzError(ERR_BLOCK_SIZE, "Invalid block size");
        Payload p;
        p.data.assign((size_t)((read + 7) >> 3) + 8, 0);
        ibs.readArray(p.data.data(), read);
        p.data.resize((size_t)((read + 7) >> 3));
        payloads.push_back(std::move(p));
    }
    size_t nblocks = payloads.size();
    std::vector<std::vector<uint8_t>> outs(nblocks);
    std::atomic<size_t> next(0);
    std::atomic<int> errCode(0);
    std::string errMsg;
    auto worker = [&]() {
        tlsBlockSize = (uint32_t)h.blockSize; tlsEntropyType = h.entropyType;   // ctx["blockSize"], ctx["entropy"] (:1385,:1406)
        for (;;) {
            size_t b = next.fetch_add(1);
            if (b >= nblocks || errCode.load()) return;
            try {
                // :1620-1626,:1649-1653 blockLength and buffer size handed to decodingTask
                size_t blk = (size_t)h.blockSize;
                blk += (512 >= (blk >> 4)) ? 512 : (blk >> 4);
                outs[b].resize(blk);
                size_t d = decodeBlock(payloads[b].data.data(), payloads[b].data.size(), h.transformType, h.entropyType,
                                       h.checksumBits, blk, outs[b].data(), outs[b].size());
                // :1707-1710 a block that decodes to more than the stream block size is rejected by the reader
                if (d > (size_t)h.blockSize) throw KnzError(ERR_PROCESS_BLOCK, "Block incorrectly decompressed");
                outs[b].resize(d);
            } catch (const KnzError& e) { int z = 0; if (errCode.compare_exchange_strong(z, e.code)) errMsg = e.what(); return; }
        }
    };
    if (jobs <= 1) worker();
    else { std::vector<std::thread> th; for (int j = 0; j < jobs; j++) th.emplace_back(worker); for (auto& t : th) t.join(); }
    if (errCode.load()) throw KnzError(errCode.load(), errMsg);
    out.clear();
    for (auto& o : outs) out.insert(out.end(), o.begin(), o.end());
}

} // namespace knzo
