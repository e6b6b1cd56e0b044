// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// CPU restatement of the kanzi-go byte transforms on the hot path.
//   v2/transform/NullTransform.go:41-77
//   v2/transform/BWTBlockCodec.go:78-137 Forward ; :141-225 Inverse ; :228 MaxEncodedLen
//   v2/transform/BWT.go:132-175 Forward ; :178-208 Inverse ; :211-358 inverseMergeTPSI ; :631-637 GetBWTChunks
//   v2/transform/DivSufSort.go:179-311 ComputeBWT/constructBWT: NOT restated. The BWT is a pure
//     function of the input (sorted suffixes, shorter-is-smaller), so the oracle builds the suffix
//     array with SA-IS and applies the output rule of :187-197 and the primary-index rule of
//     :202-206,:227-229,:283-285,:298-300,:309 (primaryIndex(k) = rank(suffix k*step)+1).
//   v2/transform/SBRT.go:127-175 Forward ; :180-226 Inverse (modes MTF/RANK/TIMESTAMP :59-76)
//   v2/transform/ZRLT.go:58-137 Forward ; :142-225 Inverse
//   v2/transform/LZCodec.go:193-236 emitLengthLZ/readLengthLZ ; :238-246 hash ; :249-591 Forward (LZ and LZX)
//     :593-607 findMatchLZX ; :621-778 inverseV6 ; :935-941 MaxEncodedLen
//   v2/transform/LZCodec.go:982-1088 LZPCodec.Forward ; :1091-1190 Inverse ; :1192-1207 findMatch ; :1210-1216 MaxEncodedLen
//   v2/transform/UTFCodec.go:84-265 Forward ; :268-383 Inverse ; :393-519 validateUTF ; :521-609 packUTF / unpackUTF1
//   v2/transform/SRT.go:49-132 Forward ; :134-167 preprocess ; :172-259 Inverse ; :261-275 encodeHeader ; :277-312 decodeHeader
#pragma once
#include "entropy_utils.hpp"
#include "divsufsort.hpp"
#include <algorithm>
#include <atomic>

namespace knzo {

struct SkipTransform : std::runtime_error { // a Forward "error" = transform not applied (Sequence.go:86-91)
    explicit SkipTransform(const std::string& m) : std::runtime_error(m) {}
};

} // namespace knzo
#include "text.hpp"      // the TEXT transform, the data types and the ctx thread-locals
namespace knzo {

// ---------------------------------------------------------------------------------------------
// Suffix array by induced sorting (SA-IS). Standard published algorithm (Nong, Zhang, Chan 2009).
// Order: plain lexicographic with "shorter is smaller" (as if a unique smallest sentinel followed).
static void saisRec(const int32_t* s, int32_t n, int32_t upper, std::vector<int32_t>& sa) {
    sa.assign((size_t)n, 0);
    if (n == 0) return;
    if (n == 1) { sa[0] = 0; return; }
    if (n == 2) { if (s[0] < s[1]) { sa[0] = 0; sa[1] = 1; } else { sa[0] = 1; sa[1] = 0; } return; }
    std::vector<uint8_t> ls((size_t)n, 0);
    for (int32_t i = n - 2; i >= 0; i--) ls[i] = (s[i] == s[i + 1]) ? ls[i + 1] : (uint8_t)(s[i] < s[i + 1]);
    std::vector<int32_t> sumL((size_t)upper + 1, 0), sumS((size_t)upper + 1, 0);
    for (int32_t i = 0; i < n; i++) {
        if (!ls[i]) sumS[s[i]]++;
        else sumL[s[i] + 1]++; // an S-type suffix has a strictly larger successor, so s[i] < upper
    }
    for (int32_t i = 0; i <= upper; i++) {
        sumS[i] += sumL[i];
        if (i < upper) sumL[i + 1] += sumS[i];
    }
    std::vector<int32_t> buf((size_t)upper + 1);
    auto induce = [&](const std::vector<int32_t>& lms) {
        std::fill(sa.begin(), sa.end(), -1);
        std::copy(sumS.begin(), sumS.end(), buf.begin());
        for (int32_t d : lms) { if (d == n) continue; sa[buf[s[d]]++] = d; }
        std::copy(sumL.begin(), sumL.end(), buf.begin());
        sa[buf[s[n - 1]]++] = n - 1;
        for (int32_t i = 0; i < n; i++) {
            int32_t v = sa[i];
            if (v >= 1 && !ls[v - 1]) sa[buf[s[v - 1]]++] = v - 1;
        }
        std::copy(sumL.begin(), sumL.end(), buf.begin());
        for (int32_t i = n - 1; i >= 0; i--) {
            int32_t v = sa[i];
            if (v >= 1 && ls[v - 1]) sa[--buf[s[v - 1] + 1]] = v - 1;
        }
    };
    std::vector<int32_t> lmsMap((size_t)n + 1, -1);
    int32_t m = 0;
    for (int32_t i = 1; i < n; i++) if (!ls[i - 1] && ls[i]) lmsMap[i] = m++;
    std::vector<int32_t> lms; lms.reserve((size_t)m);
    for (int32_t i = 1; i < n; i++) if (!ls[i - 1] && ls[i]) lms.push_back(i);
    induce(lms);
    if (m) {
        std::vector<int32_t> sortedLms; sortedLms.reserve((size_t)m);
        for (int32_t v : sa) if (lmsMap[v] != -1) sortedLms.push_back(v);
        std::vector<int32_t> recS((size_t)m);
        int32_t recUpper = 0;
        recS[lmsMap[sortedLms[0]]] = 0;
        for (int32_t i = 1; i < m; i++) {
            int32_t l = sortedLms[i - 1], r = sortedLms[i];
            int32_t endL = (lmsMap[l] + 1 < m) ? lms[lmsMap[l] + 1] : n;
            int32_t endR = (lmsMap[r] + 1 < m) ? lms[lmsMap[r] + 1] : n;
            bool same = true;
            if (endL - l != endR - r) same = false;
            else {
                while (l < endL) { if (s[l] != s[r]) break; l++; r++; }
                if (l == n || s[l] != s[r]) same = false;
            }
            if (!same) recUpper++;
            recS[lmsMap[sortedLms[i]]] = recUpper;
        }
        std::vector<int32_t> recSa;
        saisRec(recS.data(), m, recUpper, recSa);
        for (int32_t i = 0; i < m; i++) sortedLms[i] = lms[recSa[i]];
        induce(sortedLms);
    }
}

static inline void suffixArray(const uint8_t* src, int32_t n, std::vector<int32_t>& sa) {
    std::vector<int32_t> s((size_t)n);
    for (int32_t i = 0; i < n; i++) s[i] = src[i];
    saisRec(s.data(), n, 255, sa);
}

// BWT.go:631-637
static inline int getBWTChunks(int size) { return size < 256 ? 1 : 8; }

// which suffix sort BWT.forward uses: 1 = the restated DivSufSort (the reference's own algorithm, DivSufSort.go: what the CPU
// baseline times), 0 = SA-IS (independent cross-check; tests compare the two)
inline std::atomic<int>& bwtAlgo() { static std::atomic<int> a{1}; return a; }

struct BWT {
    uint64_t primaryIndexes[8] = {0};

    // BWT.go:132-175 + DivSufSort.go:179-197 (output rule) and primary index rule
    void forward(const uint8_t* src, uint8_t* dst, int count) {
        if (count == 0) return;
        if (count == 1) { dst[0] = src[0]; return; }
        if (bwtAlgo().load() == 1) {                                  // BWT.go:170: this.saAlgo.ComputeBWT(src, dst, buffer, primaryIndexes, GetBWTChunks(count))
            std::vector<int32_t> work((size_t)count + 1);
            uint32_t idx[8] = {0};
            DivSufSort d;
            d.computeBWT(src, dst, work.data(), count, idx, getBWTChunks(count));
            for (int i = 0; i < 8; i++) primaryIndexes[i] = idx[i];
            return;
        }
        std::vector<int32_t> sa;
        suffixArray(src, count, sa);
        int chunks = getBWTChunks(count);
        int32_t step = count / chunks;
        if (step * chunks != count) step++;
        dst[0] = src[count - 1];
        int32_t pIdx = -1;
        for (int32_t r = 0; r < count; r++) {
            int32_t sfx = sa[r];
            if (sfx % step == 0 && sfx / step < 8) primaryIndexes[sfx / step] = (uint64_t)(r + 1);
            if (sfx == 0) { pIdx = r; continue; }
            if (pIdx < 0) dst[r + 1] = src[sfx - 1];
            else dst[r] = src[sfx - 1];
        }
        primaryIndexes[0] = (uint64_t)(pIdx + 1);
    }

    // BWT.go:178-208 dispatch. Both reference inverses (mergeTPSI <= 4 MiB, biPSIv2 above) compute
    // the same function: follow the LF-derived successor links from primaryIndex(k)-1 for each of the
    // (1|8) chunks. Restated once with 64-bit packed links (the reference packs (next<<8|byte) in int32
    // for blocks < 2^24, BWT.go:228-247).
    void inverse(const uint8_t* src, uint8_t* dst, int count) {
        if (count == 0) return;
        if (count == 1) { dst[0] = src[0]; return; }
        int64_t pIdx = (int64_t)primaryIndexes[0];
        if (pIdx <= 0 || pIdx > count) throw KnzError(ERR_PROCESS_BLOCK, "Invalid input: corrupted BWT primary index");
        std::vector<int64_t> data((size_t)count);
        int64_t buckets[256] = {0};
        histogramO0(src, (size_t)count, buckets);
        int64_t sum = 0;
        for (int i = 0; i < 256; i++) { int64_t t = buckets[i]; buckets[i] = sum; sum += t; }
        data[buckets[src[0]]] = ((int64_t)0xFF00) | src[0];   // link of the last text symbol is never followed
        buckets[src[0]]++;
        for (int64_t i = 1; i < pIdx; i++) { int64_t v = src[i]; data[buckets[v]] = ((i - 1) << 8) | v; buckets[v]++; }
        for (int64_t i = pIdx; i < count; i++) { int64_t v = src[i]; data[buckets[v]] = (i << 8) | v; buckets[v]++; }

        if (getBWTChunks(count) != 8) {
            int64_t t = pIdx - 1;
            for (int i = 0; i < count; i++) {
                if (t < 0 || t >= count) throw KnzError(ERR_PROCESS_BLOCK, "BWT inverse: corrupted link");
                int64_t ptr = data[t]; dst[i] = (uint8_t)ptr; t = ptr >> 8;
            }
        } else {
            int64_t ckSize = count >> 3;
            if (ckSize * 8 != count) ckSize++;
            for (int c = 0; c < 8; c++) {
                int64_t t = (int64_t)primaryIndexes[c] - 1;
                if (t < 0 || t >= count) throw KnzError(ERR_PROCESS_BLOCK, "BWT inverse transform failed: corrupted BWT primary index");
                int64_t start = c * ckSize;
                int64_t end = std::min<int64_t>(start + ckSize, count);
                for (int64_t i = start; i < end; i++) {
                    if (t < 0 || t >= count) throw KnzError(ERR_PROCESS_BLOCK, "BWT inverse: corrupted link");
                    int64_t ptr = data[t]; dst[i] = (uint8_t)ptr; t = ptr >> 8;
                }
            }
        }
    }
};

static const int BWT_MAX_HEADER_SIZE = 1 + 8 * 4;

// BWTBlockCodec.go:78-137 ; returns bytes written
static inline size_t bwtBlockForward(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {
    if (n == 0 || dstCap == 0) return 0;
    if (dstCap < n + BWT_MAX_HEADER_SIZE) throw SkipTransform("Output buffer is too small");
    int blockSize = (int)n;
    uint32_t logBlockSize = log2NoCheck((uint32_t)blockSize);
    if (blockSize & (blockSize - 1)) logBlockSize++;
    int pIndexSize = (int)(logBlockSize + 7) >> 3;
    if (pIndexSize <= 0 || pIndexSize >= 5) throw SkipTransform("BWT forward failed: invalid index size");
    int chunks = getBWTChunks(blockSize);
    uint32_t logNbChunks = log2NoCheck((uint32_t)chunks);
    int headerSize = chunks * pIndexSize + 1;
    BWT bwt;
    bwt.forward(src, dst + headerSize, blockSize);
    uint8_t mode = (uint8_t)((int)(logNbChunks << 2) | (pIndexSize - 1));
    for (int i = 0, idx = 1; i < chunks; i++) {
        uint64_t primaryIndex = bwt.primaryIndexes[i] - 1;
        int shift = (pIndexSize - 1) << 3;
        while (shift >= 0) { dst[idx++] = (uint8_t)(primaryIndex >> shift); shift -= 8; }
    }
    dst[0] = mode;
    return n + (size_t)headerSize;
}

// BWTBlockCodec.go:141-225 (bsVersion 6)
static inline size_t bwtBlockInverse(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {
    if (n == 0 || dstCap == 0) return 0;
    if (n == 1) throw KnzError(ERR_PROCESS_BLOCK, "BWT inverse transform failed: invalid size");
    int blockSize = (int)n;
    uint8_t mode = src[0];
    unsigned logNbChunks = (unsigned)(mode >> 2) & 0x07;
    int pIndexSize = (int)(mode & 0x03) + 1;
    int chunks = 1 << logNbChunks;
    int headerSize = chunks * pIndexSize + 1;
    if ((int)n < headerSize || blockSize < headerSize) throw KnzError(ERR_PROCESS_BLOCK, "BWT inverse transform failed: invalid header size");
    if (chunks != getBWTChunks(blockSize - headerSize)) throw KnzError(ERR_PROCESS_BLOCK, "BWT inverse transform failed: invalid number of chunks");
    BWT bwt;
    for (int i = 0, idx = 1; i < chunks; i++) {
        int shift = (pIndexSize - 1) << 3;
        uint64_t primaryIndex = 0;
        while (shift >= 0) { primaryIndex = (primaryIndex << 8) | src[idx++]; shift -= 8; }
        if (i >= 8) throw KnzError(ERR_PROCESS_BLOCK, "BWT inverse transform failed: invalid primary index in bitstream");
        bwt.primaryIndexes[i] = primaryIndex + 1;
    }
    blockSize -= headerSize;
    if ((size_t)blockSize > dstCap) throw KnzError(ERR_PROCESS_BLOCK, "BWT inverse transform failed: output buffer too small");
    bwt.inverse(src + headerSize, dst, blockSize);
    return (size_t)blockSize;
}

// ---------------------------------------------------------------------------------------------
// SBRT.go ; mode 1 = MTF, 2 = RANK, 3 = TIMESTAMP
struct SBRT {
    int64_t mask1, mask2; unsigned shift;
    explicit SBRT(int mode) {
        mask1 = (mode == 3) ? 0 : -1;
        mask2 = (mode == 1) ? 0 : -1;
        shift = (mode == 2) ? 1 : 0;
    }
    // :127-175
    size_t forward(const uint8_t* src, size_t count, uint8_t* dst, size_t dstCap) const {
        if (count == 0 || dstCap == 0) return 0;
        if (dstCap < count + BWT_MAX_HEADER_SIZE) throw SkipTransform("SBRT forward transform skip: output buffer is too small");
        uint8_t s2r[256], r2s[256];
        for (int i = 0; i < 256; i++) { s2r[i] = (uint8_t)i; r2s[i] = (uint8_t)i; }
        int64_t p[256] = {0}, q[256] = {0};
        for (int64_t i = 0; i < (int64_t)count; i++) {
            uint8_t c = src[i];
            uint8_t r = s2r[c];
            dst[i] = r;
            int64_t qc = ((i & mask1) + (p[c] & mask2)) >> shift;
            p[c] = i;
            q[c] = qc;
            while (r > 0 && q[r2s[r - 1]] <= qc) {
                uint8_t t = r2s[r - 1];
                r2s[r] = t; s2r[t] = r;
                r--;
            }
            r2s[r] = c;
            s2r[c] = r;
        }
        return count;
    }
    // :180-226
    size_t inverse(const uint8_t* src, size_t count, uint8_t* dst, size_t dstCap) const {
        if (count == 0 || dstCap == 0) return 0;
        if (count > dstCap) throw KnzError(ERR_PROCESS_BLOCK, "SBRT inverse transform failed: output buffer too small");
        uint8_t r2s[256];
        for (int i = 0; i < 256; i++) r2s[i] = (uint8_t)i;
        int64_t p[256] = {0}, q[256] = {0};
        for (int64_t i = 0; i < (int64_t)count; i++) {
            uint8_t r = src[i];
            uint8_t c = r2s[r];
            dst[i] = c;
            int64_t qc = ((i & mask1) + (p[c] & mask2)) >> shift;
            p[c] = i;
            q[c] = qc;
            while (r > 0 && q[r2s[r - 1]] <= qc) { r2s[r] = r2s[r - 1]; r--; }
            r2s[r] = c;
        }
        return count;
    }
};

// ---------------------------------------------------------------------------------------------
// ZRLT.go:58-137. Throws SkipTransform when the output would reach len(src).
static inline size_t zrltForward(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {
    if (n == 0 || dstCap == 0) return 0;
    if (dstCap < n) throw SkipTransform("Output buffer is too small");
    uint64_t srcEnd = n, dstEnd = n;
    uint64_t srcIdx = 0, dstIdx = 0;
    bool res = true;
    while (srcIdx < srcEnd) {
        if (src[srcIdx] == 0) {
            uint64_t runStart = srcIdx - 1; // wraps at 0, as in the reference (:78)
            srcIdx++;
            while (srcIdx + 1 < srcEnd && (src[srcIdx] | src[srcIdx + 1]) == 0) srcIdx += 2;
            while (srcIdx < srcEnd && src[srcIdx] == 0) srcIdx++;
            uint64_t runLength = srcIdx - runStart;
            uint32_t lg = log2NoCheck((uint32_t)runLength);
            if (dstIdx >= dstEnd - (uint64_t)lg) { res = false; break; }
            while (lg > 0) { lg--; dst[dstIdx++] = (uint8_t)((runLength >> lg) & 1); }
            continue;
        }
        if (src[srcIdx] >= 0xFE) {
            if (dstIdx >= dstEnd - 1) { res = false; break; }
            dst[dstIdx++] = 0xFF;
            dst[dstIdx] = (uint8_t)(src[srcIdx] - 0xFE);
        } else {
            if (dstIdx >= dstEnd) { res = false; break; }
            dst[dstIdx] = (uint8_t)(src[srcIdx] + 1);
        }
        srcIdx++;
        dstIdx++;
    }
    if (srcIdx != srcEnd || !res) throw SkipTransform("ZRLT forward transform failed: output buffer is too small");
    return (size_t)dstIdx;
}

// ZRLT.go:142-225
static inline size_t zrltInverse(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {
    if (n == 0 || dstCap == 0) return 0;
    uint64_t srcEnd = n, dstEnd = dstCap;
    uint64_t srcIdx = 0, dstIdx = 0;
    uint64_t runLength = 0;
    bool err = false;
    for (;;) {
        if (src[srcIdx] <= 1) {
            runLength = 1;
            bool atEnd = false;
            while (src[srcIdx] <= 1) {
                runLength += runLength + src[srcIdx];
                srcIdx++;
                if (srcIdx >= srcEnd) { atEnd = true; break; }
            }
            if (atEnd) goto End;
            runLength--;
            if (runLength >= dstEnd - dstIdx) break;
            while (runLength > 0) { runLength--; dst[dstIdx++] = 0; }
        }
        if (src[srcIdx] == 0xFF) {
            srcIdx++;
            if (srcIdx >= srcEnd) break;
            dst[dstIdx] = (uint8_t)(0xFE + src[srcIdx]);
        } else {
            dst[dstIdx] = (uint8_t)(src[srcIdx] - 1);
        }
        srcIdx++;
        dstIdx++;
        if (srcIdx >= srcEnd || dstIdx >= dstEnd) break;
    }
End:
    if (runLength > 0) {
        runLength--;
        if (runLength > dstEnd - dstIdx) err = true;
        else while (runLength > 0) { runLength--; dst[dstIdx++] = 0; }
    }
    if (srcIdx < srcEnd) err = true;
    if (err) throw KnzError(ERR_PROCESS_BLOCK, "ZRLT inverse transform failed: output buffer is too small");
    return (size_t)dstIdx;
}

// ---------------------------------------------------------------------------------------------
// LZCodec.go (LZXCodec; extra=false is "LZ", extra=true is "LZX")
static const int LZX_MAX_DISTANCE1 = (1 << 16) - 2;
static const int LZX_MAX_DISTANCE2 = (1 << 24) - 2;
static const int LZX_MAX_MATCH = 65535 + 254 + 4;
static const int LZX_MIN_BLOCK_LENGTH = 24;

static inline size_t lzMaxEncodedLen(size_t n) { return n <= 1024 ? n + 16 : n + n / 64; }

static inline int lzEmitLength(uint8_t* block, int length) { // :193-214
    if (length < 254) { block[0] = (uint8_t)length; return 1; }
    if (length < 65536 + 254) {
        length -= 254;
        block[0] = 254; block[1] = (uint8_t)(length >> 8); block[2] = (uint8_t)length;
        return 3;
    }
    length -= 255;
    block[0] = 255; block[1] = (uint8_t)(length >> 16); block[2] = (uint8_t)(length >> 8); block[3] = (uint8_t)length;
    return 4;
}

static inline int lzReadLength(const uint8_t* block, size_t avail, int& adv) { // :216-232
    if (avail < 1) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
    int res = block[0];
    if (res < 254) { adv = 1; return res; }
    if (res == 254) {
        if (avail < 3) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
        res += (int)block[1] << 8; res += block[2]; adv = 3; return res;
    }
    if (avail < 4) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
    res += (int)block[1] << 16; res += (int)block[2] << 8; res += block[3]; adv = 4; return res;
}

static inline uint64_t le64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t le32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

static inline int lzFindMatch(const uint8_t* src, int srcIdx, int ref, int maxMatch) { // :593-607
    int bestLen = 0;
    while (bestLen + 8 <= maxMatch) {
        uint64_t diff = le64(src + srcIdx + bestLen) ^ le64(src + ref + bestLen);
        if (diff != 0) { bestLen += (__builtin_ctzll(diff) >> 3); break; }
        bestLen += 8;
    }
    return bestLen;
}

// :249-591. ctx["dataType"]: DT_DNA (min match 6) and DT_SMALL_ALPHABET (skip) are only ever set by a TEXT stage in front
static inline size_t lzForward(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap, bool extra) {
    if (n == 0 || dstCap == 0) return 0;
    int count = (int)n;
    if (dstCap < lzMaxEncodedLen(n)) throw SkipTransform("LZCodec forward transform skip: output buffer is too small");
    if (count < LZX_MIN_BLOCK_LENGTH) throw SkipTransform("LZCodec forward transform skip: block too small, skip");
    const unsigned hashLog = extra ? 19 : 16;
    const unsigned rshift = 64 - hashLog;
    std::vector<int32_t> hashes((size_t)1 << hashLog, 0);
    auto hash = [&](const uint8_t* p) -> uint32_t {
        return (uint32_t)(((le64(p) << 24) * (uint64_t)0x1E35A7BD) >> rshift);
    };
    size_t minBufSize = std::max<size_t>((size_t)count / 5, 256);
    std::vector<uint8_t> mLenBuf(minBufSize), mBuf(minBufSize), tkBuf(minBufSize);
    int srcEnd = count - 16 - 2;
    int maxDist = LZX_MAX_DISTANCE2;
    dst[12] = 1;
    if (srcEnd < 4 * LZX_MAX_DISTANCE1) { maxDist = LZX_MAX_DISTANCE1; dst[12] = 0; }
    int minMatch = 4;
    if (tlsDataType == DT_DNA) minMatch = 6;                                // :298-311: ctx["dataType"] as a TEXT stage in front left it
    else if (tlsDataType == DT_SMALL_ALPHABET) throw SkipTransform("LZCodec forward transform skip: Small alphabet");
    dst[12] |= (uint8_t)(((minMatch - 2) & 0x07) << 1);
    int srcIdx = 0, dstIdx = 13, anchor = 0, mLenIdx = 0, mIdx = 0, tkIdx = 0;
    int repd[2] = {count, count};
    int repdIdx = 0;
    int srcInc = 0;

    while (srcIdx < srcEnd) {
        int bestLen = 0;
        uint32_t h0 = hash(src + srcIdx);
        int ref0 = hashes[h0];
        hashes[h0] = srcIdx;
        uint64_t p = le64(src + srcIdx);
        int srcIdx1 = srcIdx + 1;
        int maxMatch = std::min(srcEnd - srcIdx1, LZX_MAX_MATCH);
        int ref = srcIdx1 - repd[repdIdx];
        int minRef = std::max(srcIdx - maxDist, 0);

        if (ref > minRef && (uint32_t)(p >> 8) == le32(src + ref)) {
            bestLen = lzFindMatch(src, srcIdx1, ref, maxMatch);
        } else {
            ref = srcIdx1 - repd[repdIdx ^ 1];
            if (ref > minRef && (uint32_t)(p >> 8) == le32(src + ref)) bestLen = lzFindMatch(src, srcIdx1, ref, maxMatch);
        }

        if (bestLen < minMatch) {
            ref = ref0;
            bool found = false;
            if (ref > minRef && (uint32_t)p == le32(src + ref)) {
                bestLen = lzFindMatch(src, srcIdx, ref, std::min(srcEnd - srcIdx, LZX_MAX_MATCH));
                if (bestLen >= minMatch) found = true;
            }
            if (!found) {
                srcIdx = srcIdx1 + (srcInc >> 6);
                srcInc++;
                repdIdx = 0;
                continue;
            }
            // checkNext:
            if (ref != srcIdx - repd[0] && ref != srcIdx - repd[1]) {
                uint32_t h1 = hash(src + srcIdx1);
                int ref1 = hashes[h1];
                hashes[h1] = srcIdx1;
                if (ref1 > minRef + 1 && le32(src + srcIdx1 + bestLen - 3) == le32(src + ref1 + bestLen - 3)) {
                    int bestLen1 = lzFindMatch(src, srcIdx1, ref1, maxMatch);
                    if (bestLen1 >= bestLen) { ref = ref1; bestLen = bestLen1; srcIdx = srcIdx1; }
                }
                if (extra) {
                    int srcIdx2 = srcIdx1 + 1;
                    uint32_t h2 = hash(src + srcIdx2);
                    int ref2 = hashes[h2];
                    hashes[h2] = srcIdx2;
                    if (ref2 > minRef + 2 && le32(src + srcIdx2 + bestLen - 3) == le32(src + ref2 + bestLen - 3)) {
                        int bestLen2 = lzFindMatch(src, srcIdx2, ref2, std::min(srcEnd - srcIdx2, LZX_MAX_MATCH));
                        if (bestLen2 >= bestLen) { ref = ref2; bestLen = bestLen2; srcIdx = srcIdx2; }
                    }
                }
            }
            while (srcIdx > anchor && ref > minRef && src[srcIdx - 1] == src[ref - 1]) { bestLen++; ref--; srcIdx--; }
            if (bestLen > LZX_MAX_MATCH) {
                srcIdx += (bestLen - LZX_MAX_MATCH);
                ref += (bestLen - LZX_MAX_MATCH);
                bestLen = LZX_MAX_MATCH;
            }
        } else {
            if (src[srcIdx] == src[ref - 1] && bestLen < LZX_MAX_MATCH) {
                bestLen++;
                ref--;
            } else {
                srcIdx++;
                uint32_t h1 = hash(src + srcIdx);
                hashes[h1] = srcIdx;
            }
        }

        srcInc = 0;
        int dist = srcIdx - ref;
        int mLen = bestLen - minMatch;
        int token, mLenTh;
        if (dist == repd[0]) { token = 0x00; mLenTh = 3; }
        else if (dist == repd[1]) { token = 0x04; mLenTh = 3; }
        else {
            mLenTh = 7;
            if (dist >= 256) {
                if (dist >= 65536) {
                    mBuf[mIdx] = (uint8_t)(dist >> 16); mBuf[mIdx + 1] = (uint8_t)(dist >> 8); mIdx += 2; token = 0x18;
                } else {
                    mBuf[mIdx] = (uint8_t)(dist >> 8); mIdx++; token = 0x10;
                }
            } else token = 0x08;
            mBuf[mIdx++] = (uint8_t)dist;
        }
        if (mLen >= mLenTh) { token += mLenTh; mLenIdx += lzEmitLength(&mLenBuf[mLenIdx], mLen - mLenTh); }
        else token += mLen;
        repd[1] = repd[0];
        repd[0] = dist;
        repdIdx = 1;
        int litLen = srcIdx - anchor;
        if (tkIdx >= (int)tkBuf.size()) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); // Go never grows tkBuf
        if (litLen == 0) {
            tkBuf[tkIdx++] = (uint8_t)token;
        } else {
            if (litLen >= 7) {
                if (litLen >= (1 << 24)) throw SkipTransform("LZCodec forward transform skip: too many literals");
                tkBuf[tkIdx++] = (uint8_t)((7 << 5) | token);
                dstIdx += lzEmitLength(dst + dstIdx, litLen - 7);
            } else {
                tkBuf[tkIdx++] = (uint8_t)((litLen << 5) | token);
            }
            memcpy(dst + dstIdx, src + anchor, (size_t)litLen);
            dstIdx += litLen;
        }
        if (mIdx >= (int)mBuf.size() - 8) {
            mBuf.resize(mBuf.size() + mBuf.size() / 2);
            if (mLenIdx >= (int)mLenBuf.size() - 8) mLenBuf.resize(mLenBuf.size() + mLenBuf.size() / 2);
        }
        anchor = srcIdx + bestLen;
        while (srcIdx + 4 < anchor) { // :532-543 (unrolled by 4 in the reference; same inserts, same order)
            srcIdx += 4;
            uint64_t v = le64(src + srcIdx - 3);
            uint32_t a0 = (uint32_t)((((v >> 0) << 24) * (uint64_t)0x1E35A7BD) >> rshift);
            uint32_t a1 = (uint32_t)((((v >> 8) << 24) * (uint64_t)0x1E35A7BD) >> rshift);
            uint32_t a2 = (uint32_t)((((v >> 16) << 24) * (uint64_t)0x1E35A7BD) >> rshift);
            uint32_t a3 = (uint32_t)((((v >> 24) << 24) * (uint64_t)0x1E35A7BD) >> rshift);
            hashes[a0] = srcIdx - 3; hashes[a1] = srcIdx - 2; hashes[a2] = srcIdx - 1; hashes[a3] = srcIdx;
        }
        srcIdx++;
        while (srcIdx < anchor) { hashes[hash(src + srcIdx)] = srcIdx; srcIdx++; }
    }

    int litLen = count - anchor;
    if (dstIdx + litLen + tkIdx + mIdx >= count) throw SkipTransform("LZCodec forward transform skip: no compression");
    if (tkIdx >= (int)tkBuf.size()) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
    if (litLen >= 7) {
        tkBuf[tkIdx++] = (uint8_t)(7 << 5);
        dstIdx += lzEmitLength(dst + dstIdx, litLen - 7);
    } else {
        tkBuf[tkIdx++] = (uint8_t)(litLen << 5);
    }
    memcpy(dst + dstIdx, src + anchor, (size_t)litLen);
    dstIdx += litLen;
    uint32_t u;
    u = (uint32_t)dstIdx; memcpy(dst + 0, &u, 4);
    u = (uint32_t)tkIdx;  memcpy(dst + 4, &u, 4);
    u = (uint32_t)mIdx;   memcpy(dst + 8, &u, 4);
    if ((size_t)dstIdx + tkIdx + mIdx + mLenIdx > dstCap) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
    memcpy(dst + dstIdx, tkBuf.data(), (size_t)tkIdx); dstIdx += tkIdx;
    memcpy(dst + dstIdx, mBuf.data(), (size_t)mIdx); dstIdx += mIdx;
    memcpy(dst + dstIdx, mLenBuf.data(), (size_t)mLenIdx); dstIdx += mLenIdx;
    if (dstIdx > count - count / 100) throw SkipTransform("LZCodec forward transform skip: no compression");
    return (size_t)dstIdx;
}

// :621-778
static inline size_t lzInverse(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {
    if (n == 0 || dstCap == 0) return 0;
    int64_t count = (int64_t)n;
    if (count < 13) throw KnzError(ERR_PROCESS_BLOCK, "LZCodec inverse transform failed: invalid data");
    // Go reads these as uint32 -> int (64-bit), so they are never negative
    int64_t tkIdx = (int64_t)le32(src), mIdx = (int64_t)le32(src + 4), mLenIdx = (int64_t)le32(src + 8);
    mIdx += tkIdx;
    mLenIdx += mIdx;
    if (tkIdx > count || mIdx > count || mLenIdx > count) throw KnzError(ERR_PROCESS_BLOCK, "LZCodec inverse transform failed: invalid data");
    int64_t srcEnd = tkIdx - 13;
    int mFlag = src[12] & 0x01;
    int64_t dstEnd = (int64_t)dstCap - 16;
    int64_t maxDist = mFlag == 0 ? LZX_MAX_DISTANCE1 : LZX_MAX_DISTANCE2;
    int minMatch = ((src[12] >> 1) & 0x07) + 2;
    int64_t srcIdx = 13, dstIdx = 0;
    int64_t repd0 = count, repd1 = count;
    auto need = [&](int64_t idx) { if (idx < 0 || idx >= count) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); };

    for (;;) {
        need(tkIdx);
        int token = src[tkIdx++];
        if (token >= 32) {
            int64_t litLen;
            if (token >= 0xE0) {
                int adv; need(srcIdx);
                int ll = lzReadLength(src + srcIdx, (size_t)(count - srcIdx), adv);
                litLen = 7 + ll; srcIdx += adv;
            } else litLen = token >> 5;
            if (srcIdx + litLen > count || dstIdx + litLen > (int64_t)dstCap) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
            memcpy(dst + dstIdx, src + srcIdx, (size_t)litLen);
            srcIdx += litLen;
            dstIdx += litLen;
            if (srcIdx >= srcEnd) break;
        }
        int64_t mLen, dist;
        int f = token & 0x18;
        if (f == 0) {
            mLen = token & 0x03;
            if (mLen == 3) { int adv; need(mLenIdx); int ml = lzReadLength(src + mLenIdx, (size_t)(count - mLenIdx), adv); mLen += minMatch + ml; mLenIdx += adv; }
            else mLen += minMatch;
            dist = (token & 0x04) == 0 ? repd0 : repd1;
        } else {
            mLen = token & 0x07;
            if (mLen == 7) { int adv; need(mLenIdx); int ml = lzReadLength(src + mLenIdx, (size_t)(count - mLenIdx), adv); mLen += minMatch + ml; mLenIdx += adv; }
            else mLen += minMatch;
            need(mIdx); dist = src[mIdx++];
            if (f >= 0x10) {
                need(mIdx); dist = (dist << 8) | src[mIdx++];
                if (f == 0x18) { need(mIdx); dist = (dist << 8) | src[mIdx++]; }
            }
        }
        repd1 = repd0;
        repd0 = dist;
        int64_t mEnd = dstIdx + mLen;
        int64_t ref = dstIdx - dist;
        if (ref < 0 || dist > maxDist || mEnd > dstEnd) throw KnzError(ERR_PROCESS_BLOCK, "LZCodec: invalid distance decoded");
        for (int64_t i = 0; i < mLen; i++) dst[dstIdx + i] = dst[ref + i];
        dstIdx = mEnd;
    }
    if (srcIdx != srcEnd + 13) throw KnzError(ERR_PROCESS_BLOCK, "LZCodec inverse transform failed");
    return (size_t)dstIdx;
}

// ---------------------------------------------------------------------------------------------
// LZP (LZCodec.go:943-1216). Hash of the last 4 bytes predicts one position; a prediction that holds for >= 64 bytes is
// replaced by 0xFC + length, a literal 0xFC whose context had a prediction is escaped with 0xFF.
static const int LZP_HASH_LOG = 16;
static const uint32_t LZP_HASH_SEED = 0x7FEB352D;
static const int LZP_MIN_MATCH64 = 64;
static const int LZP_MATCH_FLAG = 0xFC;
static const int LZP_MIN_BLOCK_LENGTH = 128;

static inline size_t lzpForward(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {   // :982-1088
    if (n == 0 || dstCap == 0) return 0;
    const int count = (int)n;
    if (dstCap < lzMaxEncodedLen(n)) throw SkipTransform("Output buffer is too small");
    if (count < LZP_MIN_BLOCK_LENGTH) throw SkipTransform("Block too small, skip");
    const int srcEnd = count, dstEnd = count - (count >> 6);
    std::vector<int32_t> hashes((size_t)1 << LZP_HASH_LOG, 0);
    memcpy(dst, src, 4);
    uint32_t ctx = le32(src);
    int srcIdx = 4, dstIdx = 4;
    while (srcIdx < srcEnd - LZP_MIN_MATCH64 && dstIdx < dstEnd) {
        const uint32_t h = (LZP_HASH_SEED * ctx) >> (32 - LZP_HASH_LOG);
        const int ref = hashes[h];
        hashes[h] = srcIdx;
        int bestLen = 0;
        if (ref != 0 && le64(src + srcIdx + LZP_MIN_MATCH64 - 8) == le64(src + ref + LZP_MIN_MATCH64 - 8))
            bestLen = lzFindMatch(src, srcIdx, ref, srcEnd - srcIdx);   // :1192-1207 is the same 8-byte stride loop
        if (bestLen < LZP_MIN_MATCH64) {
            const uint32_t val = src[srcIdx];
            ctx = (ctx << 8) | val;
            dst[dstIdx++] = src[srcIdx++];
            if (ref != 0 && val == (uint32_t)LZP_MATCH_FLAG) dst[dstIdx++] = 0xFF;
            continue;
        }
        srcIdx += bestLen;
        ctx = le32(src + srcIdx - 4);
        dst[dstIdx++] = (uint8_t)LZP_MATCH_FLAG;
        bestLen -= LZP_MIN_MATCH64;
        while (bestLen >= 254) {
            bestLen -= 254;
            dst[dstIdx++] = 0xFE;
            if (dstIdx >= dstEnd) break;
        }
        dst[dstIdx++] = (uint8_t)bestLen;
    }
    while (srcIdx < srcEnd && dstIdx < dstEnd) {
        const uint32_t h = (LZP_HASH_SEED * ctx) >> (32 - LZP_HASH_LOG);
        const int ref = hashes[h];
        hashes[h] = srcIdx;
        const uint32_t val = src[srcIdx];
        ctx = (ctx << 8) | val;
        dst[dstIdx++] = src[srcIdx++];
        if (ref != 0 && val == (uint32_t)LZP_MATCH_FLAG) dst[dstIdx++] = 0xFF;
    }
    if (srcIdx != count || dstIdx >= dstEnd) throw SkipTransform("LZP forward transform skip: output buffer too small");
    return (size_t)dstIdx;
}

static inline size_t lzpInverse(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {   // :1091-1190
    if (n == 0 || dstCap == 0) return 0;
    if (n < 4) throw KnzError(ERR_PROCESS_BLOCK, "LZP inverse transform failed: block too small");
    if (dstCap < 4) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
    std::vector<int32_t> hashes((size_t)1 << LZP_HASH_LOG, 0);
    const int64_t srcEnd = (int64_t)n, dstEnd = (int64_t)dstCap;
    memcpy(dst, src, 4);
    uint32_t ctx = le32(dst);
    int64_t srcIdx = 4, dstIdx = 4;
    bool res = true;
    const int minMatch = LZP_MIN_MATCH64;                    // bitstream version >= 4
    auto needS = [&](int64_t i) { if (i < 0 || i >= srcEnd) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); };
    auto needD = [&](int64_t i) { if (i < 0 || i >= dstEnd) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); };
    while (srcIdx < srcEnd) {
        const uint32_t h = (LZP_HASH_SEED * ctx) >> (32 - LZP_HASH_LOG);
        const int64_t ref = hashes[h];
        hashes[h] = (int32_t)dstIdx;
        if (src[srcIdx] != LZP_MATCH_FLAG || ref == 0) {
            needD(dstIdx);
            dst[dstIdx] = src[srcIdx];
            ctx = (ctx << 8) | dst[dstIdx];
            srcIdx++; dstIdx++;
            continue;
        }
        srcIdx++;
        needS(srcIdx);
        if (src[srcIdx] == 0xFF) {
            needD(dstIdx);
            dst[dstIdx] = (uint8_t)LZP_MATCH_FLAG;
            ctx = (ctx << 8) | (uint32_t)LZP_MATCH_FLAG;
            srcIdx++; dstIdx++;
            continue;
        }
        int64_t mLen = minMatch;
        if (src[srcIdx] == 0xFE) {
            while (srcIdx < srcEnd && src[srcIdx] == 0xFE) { srcIdx++; mLen += 254; }
            if (srcIdx >= srcEnd) { res = false; break; }
        }
        mLen += src[srcIdx++];
        const int64_t mEnd = dstIdx + mLen;
        if (mEnd > dstEnd) { res = false; break; }
        for (int64_t i = 0; i < mLen; i++) dst[dstIdx + i] = dst[ref + i];   // (copy() when the ranges are apart: same bytes)
        dstIdx += mLen;
        ctx = le32(dst + dstIdx - 4);
    }
    if (!res || srcIdx != srcEnd) throw KnzError(ERR_PROCESS_BLOCK, "LZP inverse transform failed: output buffer too small");
    return (size_t)dstIdx;
}

// ---------------------------------------------------------------------------------------------
// SRT, sorted rank transform (SRT.go). Move-to-front ranks of the run heads, written per symbol (bucket of symbol c =
// the ranks of c's occurrences, buckets ordered by decreasing frequency), behind a header of 256 varint frequencies.
static const int SRT_MAX_HEADER_SIZE = 4 * 256;

static inline int srtPreprocess(const int32_t* freqs, uint8_t* symbols) {   // :134-167 (shell sort, decreasing frequency, ties by symbol)
    int nbSymbols = 0;
    for (int i = 0; i < 256; i++) if (freqs[i] != 0) symbols[nbSymbols++] = (uint8_t)i;
    int h = 4;
    while (h < nbSymbols) h = h * 3 + 1;
    for (;;) {
        h /= 3;
        for (int i = h; i < nbSymbols; i++) {
            const uint8_t t = symbols[i];
            int b;
            for (b = i - h; b >= 0 && (freqs[symbols[b]] < freqs[t] || (t < symbols[b] && freqs[t] == freqs[symbols[b]])); b -= h)
                symbols[b + h] = symbols[b];
            symbols[b + h] = t;
        }
        if (h == 1) break;
    }
    return nbSymbols;
}

static inline size_t srtForward(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {   // :49-132
    if (n == 0 || dstCap == 0) return 0;
    if (dstCap < n + SRT_MAX_HEADER_SIZE) throw SkipTransform("Output buffer is too small");
    const int count = (int)n;
    uint8_t s2r[256] = {0}, r2s[256] = {0};
    int32_t freqs[256] = {0};
    for (int i = 0, b = 0; i < count;) {
        const uint8_t c = src[i];
        if (freqs[c] == 0) { r2s[b] = c; s2r[c] = (uint8_t)b; b++; }
        int j = i + 1;
        while (j < count && src[j] == c) j++;
        freqs[c] += j - i;
        i = j;
    }
    uint8_t symbols[256] = {0};
    const int nbSymbols = srtPreprocess(freqs, symbols);
    int buckets[256] = {0};
    for (int i = 0, pos = 0; i < nbSymbols; i++) { const uint8_t c = symbols[i]; buckets[c] = pos; pos += freqs[c]; }
    int hs = 0;
    for (int i = 0; i < 256; i++) {                           // encodeHeader :261-275
        int32_t f = freqs[i];
        while (f >= 128) { dst[hs++] = (uint8_t)(0x80 | (f & 0x7F)); f >>= 7; }
        dst[hs++] = (uint8_t)f;
    }
    uint8_t* out = dst + hs;
    for (int i = 0; i < count;) {
        const uint8_t c = src[i];
        uint8_t r = s2r[c];
        int p = buckets[c];
        out[p++] = r;
        if (r > 0) {
            for (;;) {
                const uint8_t t = r2s[r - 1];
                r2s[r] = t; s2r[t] = r;
                if (r == 1) break;
                r--;
            }
            r2s[0] = c; s2r[c] = 0;
        }
        i++;
        while (i < count && src[i] == c) { out[p++] = 0; i++; }
        buckets[c] = p;
    }
    return (size_t)(count + hs);
}

static inline size_t srtInverse(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {   // :172-259
    if (n == 0 || dstCap == 0) return 0;
    int32_t freqs[256];
    size_t hs = 0;
    auto rd = [&]() -> int32_t { if (hs >= n) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); return src[hs++]; };
    for (int i = 0; i < 256; i++) {                           // decodeHeader :277-312
        int32_t val = rd();
        if (val < 128) { freqs[i] = val; continue; }
        int32_t res = val & 0x7F;
        val = rd(); res |= (val & 0x7F) << 7;
        if (val >= 128) {
            val = rd(); res |= (val & 0x7F) << 14;
            if (val >= 128) { val = rd(); res |= (val & 0x7F) << 21; }
        }
        freqs[i] = res;
    }
    const uint8_t* in = src + hs;
    const int64_t len = (int64_t)(n - hs);
    if (len > (int64_t)dstCap) throw KnzError(ERR_PROCESS_BLOCK, "SRT inverse transform failed: invalid data");
    uint8_t symbols[256] = {0};
    int nbSymbols = srtPreprocess(freqs, symbols);
    int64_t buckets[256] = {0}, bucketEnds[256] = {0};
    uint8_t r2s[256] = {0};
    int64_t bucketPos = 0;
    for (int i = 0; i < nbSymbols; i++) {
        const uint8_t c = symbols[i];
        if (bucketPos < 0 || bucketPos > len) throw KnzError(ERR_PROCESS_BLOCK, "SRT inverse transform failed: invalid data");
        if (bucketPos >= len) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
        r2s[in[bucketPos]] = c;
        buckets[c] = bucketPos + 1;
        bucketPos += freqs[c];
        bucketEnds[c] = bucketPos;
    }
    uint8_t c = r2s[0];
    for (size_t i = 0; i < dstCap; i++) {                     // (the reference walks all of dst; only the first `len` bytes are returned)
        dst[i] = c;
        if (buckets[c] < bucketEnds[c]) {
            if (buckets[c] >= len) throw KnzError(ERR_PROCESS_BLOCK, "index out of range");
            const uint8_t r = in[buckets[c]];
            buckets[c]++;
            if (r == 0) continue;
            for (int s = 0; s < (int)r; s++) r2s[s] = r2s[s + 1];
            r2s[r] = c;
            c = r2s[0];
        } else {
            if (nbSymbols == 1) continue;
            nbSymbols--;
            for (int s = 0; s < nbSymbols; s++) r2s[s] = r2s[s + 1];
            c = r2s[0];
        }
    }
    return (size_t)len;
}

// ---------------------------------------------------------------------------------------------
// UTF codec (UTFCodec.go): UTF-8 code points -> 1- or 2-byte ranks by decreasing frequency, behind a map of the code points.
// ctx["dataType"] (internal/Global.go:26-40) travels in a thread-local here: encodeBlock sets it from the block's magic number
// (io/CompressedStream.go:811-819), a transform object used on its own sees DT_UNDEFINED.
static const int UTF_MIN_BLOCKSIZE = 1024;

static inline int utfSize(uint8_t b) {                       // _UTF_SIZES :31-48
    if (b < 0x80) return 1;
    if (b < 0xC2) return 0;
    if (b < 0xE0) return 2;
    if (b < 0xF0) return 3;
    if (b < 0xF5) return 4;
    return 0;
}

static inline bool utfValidate(const uint8_t* block, int count) {   // validateUTF :393-519
    std::vector<int> freqs0(256, 0);
    std::vector<int> freqs1(65536, 0);
    const int end4 = count & -4;
    uint8_t prv = 0;
    auto rule1 = [&]() { int sum = freqs0[0xC0] + freqs0[0xC1]; for (int k = 0xF5; k < 256; k++) sum += freqs0[k]; return sum == 0; };
    for (int i = 0; i < end4; i += 4) {
        const uint8_t c0 = block[i], c1 = block[i + 1], c2 = block[i + 2], c3 = block[i + 3];
        freqs0[c0]++; freqs0[c1]++; freqs0[c2]++; freqs0[c3]++;
        freqs1[(prv << 8) | c0]++; freqs1[(c0 << 8) | c1]++; freqs1[(c1 << 8) | c2]++; freqs1[(c2 << 8) | c3]++;
        prv = c3;
        if ((i & 0x0FFF) == 0 && !rule1()) return false;
    }
    if (end4 != count) {
        for (int i = end4; i < count; i++) { const uint8_t cur = block[i]; freqs0[cur]++; freqs1[(prv << 8) | cur]++; prv = cur; }
        if (!rule1()) return false;
    }
    int sum = 0, sum2 = 0;
    for (int i = 0; i < 256; i++) {
        if (i < 0xA0 || i > 0xBF) sum += freqs1[(0xE0 << 8) | i];
        if (i < 0x80 || i > 0x9F) sum += freqs1[(0xED << 8) | i];
        if (i < 0x90 || i > 0xBF) sum += freqs1[(0xF0 << 8) | i];
        if (i < 0x80 || i > 0x8F) sum += freqs1[(0xF4 << 8) | i];
        if (i < 0x80 || i > 0xBF) {
            for (int j = 0xC2; j <= 0xDF; j++) sum += freqs1[(j << 8) | i];
            for (int j = 0xE1; j <= 0xEC; j++) sum += freqs1[(j << 8) | i];
            sum += freqs1[(0xF1 << 8) | i] + freqs1[(0xF2 << 8) | i] + freqs1[(0xF3 << 8) | i];
            sum += freqs1[(0xEE << 8) | i] + freqs1[(0xEF << 8) | i];
        } else sum2 += freqs0[i];
        if (sum != 0) return false;
    }
    return sum2 >= count / 8;
}

static inline int utfPack(const uint8_t* in, uint32_t& out) {       // packUTF :521-546
    const int s = utfSize(in[0]);
    switch (s) {
        case 1: out = in[0]; break;
        case 2: out = (1u << 19) | ((uint32_t)in[0] << 8) | in[1]; break;
        case 3: out = (2u << 19) | (((uint32_t)in[0] & 0x0F) << 12) | (((uint32_t)in[1] & 0x3F) << 6) | ((uint32_t)in[2] & 0x3F); break;
        case 4: out = (4u << 19) | (((uint32_t)in[0] & 0x07) << 18) | (((uint32_t)in[1] & 0x3F) << 12) | (((uint32_t)in[2] & 0x3F) << 6) | ((uint32_t)in[3] & 0x3F); break;
        default: out = 0; break;
    }
    return s;
}

static inline int utfUnpack1(uint32_t in, uint8_t* out) {           // unpackUTF1 :578-609 (bitstream version >= 4)
    const uint32_t sz = in >> 19;
    if (sz == 0) { out[0] = (uint8_t)in; return 1; }
    if (sz == 1) { out[0] = (uint8_t)(in >> 8); out[1] = (uint8_t)in; return 2; }
    if (sz == 2) { out[0] = (uint8_t)(((in >> 12) & 0x0F) | 0xE0); out[1] = (uint8_t)(((in >> 6) & 0x3F) | 0x80); out[2] = (uint8_t)((in & 0x3F) | 0x80); return 3; }
    if (sz >= 4 && sz <= 7) {
        out[0] = (uint8_t)(((in >> 18) & 0x07) | 0xF0); out[1] = (uint8_t)(((in >> 12) & 0x3F) | 0x80);
        out[2] = (uint8_t)(((in >> 6) & 0x3F) | 0x80); out[3] = (uint8_t)((in & 0x3F) | 0x80);
        return 4;
    }
    return 0;
}

static inline size_t utfForward(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {   // :84-265
    if (n == 0 || dstCap == 0) return 0;
    if (n < (size_t)UTF_MIN_BLOCKSIZE) throw SkipTransform("Input block is too small");
    if (dstCap < n + 8192) throw SkipTransform("Output buffer is too small");
    const int count = (int)n;
    bool mustValidate = true;
    {
        const int dt = tlsDataType;
        if (dt != DT_UNDEFINED && dt != DT_UTF8) throw SkipTransform("UTF forward transform skip: not UTF");
        mustValidate = dt != DT_UTF8;
    }
    int start = 0;
    const uint32_t be32 = ((uint32_t)src[0] << 24) | ((uint32_t)src[1] << 16) | ((uint32_t)src[2] << 8) | src[3];
    if ((be32 & 0x00FFFFFFu) == 0x00EFBBBFu) start = 3;
    else while (start < 4 && utfSize(src[start]) == 0) start++;
    if (mustValidate && !utfValidate(src + start, count - 4 - start)) throw SkipTransform("UTF forward transform skip: not UTF");
    tlsDataType = DT_UTF8;
    std::vector<int32_t> aliasMap((size_t)1 << 22, 0);
    struct Sd { int32_t sym, freq; };
    std::vector<Sd> symb(32768);
    int nsym = 0;
    for (int i = start; i < count - 4;) {
        uint32_t val;
        const int s = utfPack(src + i, val);
        bool res = s != 0;
        res = res && (s != 3 || (src[i + 2] & 0xC0) == 0x80);
        res = res && (s != 4 || ((((uint32_t)src[i + 2] << 8) | src[i + 3]) & 0xC0C0u) == 0x8080u);
        if (aliasMap[val] == 0) {
            if (nsym < 32768) symb[nsym].sym = (int32_t)val;      // (Go would panic past the array only after res turned false: n < 32768 is tested first)
            nsym++;
            res = res && nsym < 32768;
        }
        if (!res) throw SkipTransform("UTF forward transform skip: invalid or too complex");
        aliasMap[val]++;
        i += s;
    }
    if (nsym == 0) throw SkipTransform("UTF forward transform skip: not UTF");
    const int maxTarget = count - count / 10;
    if (3 * nsym + 6 >= maxTarget) throw SkipTransform("UTF forward transform skip: no improvement");
    for (int i = 0; i < nsym; i++) symb[i].freq = aliasMap[symb[i].sym];
    std::stable_sort(symb.begin(), symb.begin() + nsym, [](const Sd& a, const Sd& b) { return a.freq != b.freq ? a.freq < b.freq : a.sym < b.sym; });
    int dstIdx = 2;
    dst[dstIdx++] = (uint8_t)(nsym >> 8);
    dst[dstIdx++] = (uint8_t)nsym;
    int64_t estimate = dstIdx + 6;
    for (int i = 0; i < nsym; i++) {
        const int r = nsym - 1 - i;
        const int32_t s = symb[r].sym;
        dst[dstIdx] = (uint8_t)(s >> 16); dst[dstIdx + 1] = (uint8_t)(s >> 8); dst[dstIdx + 2] = (uint8_t)s;
        dstIdx += 3;
        if (i < 128) { estimate += symb[r].freq; aliasMap[s] = i; }
        else { estimate += 2 * (int64_t)symb[r].freq; aliasMap[s] = 0x10080 | ((i << 1) & 0xFF00) | (i & 0x7F); }
    }
    if (estimate >= maxTarget) throw SkipTransform("UTF forward transform skip: no improvement");
    for (int i = 0; i < start; i++) dst[dstIdx++] = src[i];
    int srcIdx = start;
    while (srcIdx < count - 4) {
        uint32_t val;
        srcIdx += utfPack(src + srcIdx, val);
        const int32_t alias = aliasMap[val];
        dst[dstIdx++] = (uint8_t)alias;
        dst[dstIdx] = (uint8_t)(alias >> 8);
        dstIdx += alias >> 16;
    }
    dst[0] = (uint8_t)start;
    dst[1] = (uint8_t)(srcIdx - (count - 4));
    while (srcIdx < count) dst[dstIdx++] = src[srcIdx++];
    if (dstIdx >= maxTarget) throw SkipTransform("UTF forward transform skip: no improvement");
    return (size_t)dstIdx;
}

static inline size_t utfInverse(const uint8_t* src, size_t n, uint8_t* dst, size_t dstCap) {   // :268-383
    if (n == 0 || dstCap == 0) return 0;
    if (n < 4) throw KnzError(ERR_PROCESS_BLOCK, "Input block is too small");
    const int64_t count = (int64_t)n;
    const int start = src[0] & 0x03, adjust = src[1] & 0x03;
    const int nsym = ((int)src[2] << 8) + src[3];
    if (nsym == 0 || nsym >= 32768 || 4 + 3 * (int64_t)nsym > count) throw KnzError(ERR_PROCESS_BLOCK, "UTF inverse transform: invalid map size");
    struct Sym { uint8_t value[4]; uint8_t length; };
    std::vector<Sym> m(32768, Sym{{0, 0, 0, 0}, 0});
    int64_t srcIdx = 4;
    for (int i = 0; i < nsym; i++) {
        const uint32_t s = ((uint32_t)src[srcIdx] << 16) | ((uint32_t)src[srcIdx + 1] << 8) | src[srcIdx + 2];
        const int sl = utfUnpack1(s, m[i].value);
        if (sl == 0) throw KnzError(ERR_PROCESS_BLOCK, "UTF inverse transform failed: invalid UTF alias");
        m[i].length = (uint8_t)sl;
        srcIdx += 3;
    }
    const int64_t srcEnd = count - 4 + adjust;
    int64_t dstIdx = 0;
    const int64_t dstEnd = (int64_t)dstCap - 4;
    if (dstEnd < 0) throw KnzError(ERR_PROCESS_BLOCK, "UTF inverse transform failed: invalid output block size");
    if (srcEnd < srcIdx || srcEnd > count || srcIdx + start > count) throw KnzError(ERR_PROCESS_BLOCK, "UTF inverse transform failed: invalid data");
    for (int i = 0; i < start; i++) { if (dstIdx >= (int64_t)dstCap) throw KnzError(ERR_PROCESS_BLOCK, "index out of range"); dst[dstIdx++] = src[srcIdx++]; }
    while (srcIdx < srcEnd && dstIdx < dstEnd) {
        int alias = src[srcIdx++];
        if (alias >= 128) {
            if (srcIdx >= srcEnd) throw KnzError(ERR_PROCESS_BLOCK, "UTF inverse transform failed: invalid data");
            alias = ((int)src[srcIdx] << 7) + (alias & 0x7F);
            srcIdx++;
        }
        const Sym& s = m[alias];
        memcpy(dst + dstIdx, s.value, 4);                      // copy(dst[dstIdx:], s.value[:4]): dstIdx < dstEnd = len - 4
        dstIdx += s.length;
    }
    if (srcIdx < srcEnd || dstIdx > (int64_t)dstCap - count + srcEnd) throw KnzError(ERR_PROCESS_BLOCK, "UTF inverse transform failed: invalid data");
    for (int64_t i = srcEnd; i < count; i++) dst[dstIdx++] = src[srcIdx++];
    return (size_t)dstIdx;
}

} // namespace knzo
