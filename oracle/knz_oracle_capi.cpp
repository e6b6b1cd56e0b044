// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.
// C entry points over the CPU restatement of kanzi-go's block pipeline (see the headers in this
// directory for the reference file:line each function follows). Loaded through ctypes by tests/,
// by __graft_entry__.smoke() and by bench.py's cpu_baseline leg — never by the product library.
//
// PARITY PIN STATUS (round 5): pinned by the reference's own code. oracle/_ref = kanzi-go's .go sources translated mechanically to C++
// (tools/go2cpp) and compiled; tests/test_ref_build.py requires _ref == this oracle on every codec, transform, sequence, hash and whole
// stream, and the 387 stream vectors under tests/golden/ref_streams/ were written by the reference's own Writer (tests/test_ref_streams.py).
// Also pinned (tests/test_oracle_units.py): BWT("mississippi")="ipssmpissii"/5 (BWT.go:48-62), the signed Exp-Golomb table
// (ExpGolombCodec.go:45-62), varint lengths (Entropy_test.go:84-96), stream header layout/constants (CompressedStream.go:442-516), XXH32 known answers.
#include "stream.hpp"

using namespace knzo;

static thread_local std::string g_lastError;

#define KNZO_TRY try {
#define KNZO_CATCH                                                          \
    }                                                                       \
    catch (const KnzError& e) { g_lastError = e.what(); return e.code; }    \
    catch (const SkipTransform& e) { g_lastError = e.what(); return -1; }   \
    catch (const std::exception& e) { g_lastError = e.what(); return ERR_UNKNOWN; }

extern "C" {

const char* knzo_last_error() { return g_lastError.c_str(); }

int knzo_entropy_encode(uint32_t type, const uint8_t* src, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_bits) {
    KNZO_TRY
    BitWriter obs;
    entropyEncode(obs, type, src, (size_t)n);
    uint64_t bits = obs.close();
    if (obs.buf.size() > cap) throw KnzError(ERR_WRITE_FILE, "output buffer too small");
    if (!obs.buf.empty()) memcpy(out, obs.buf.data(), obs.buf.size());
    *out_bits = bits;
    return 0;
    KNZO_CATCH
}

int knzo_entropy_decode(uint32_t type, const uint8_t* bits, uint64_t nbytes, uint8_t* dst, uint64_t n, uint64_t* used_bits) {
    KNZO_TRY
    BitReader ibs(bits, nbytes);
    entropyDecode(ibs, type, dst, (size_t)n);
    if (used_bits) *used_bits = ibs.read();
    return 0;
    KNZO_CATCH
}

// DefaultOutputBitStream / DefaultInputBitStream on one small case: optional bit in front, WriteArray(arr, nbits), Close(), then ReadBits(read_len)
// (the shape of bitstream/DefaultBitstream_test.go:476-528, whose expected values pin the bit order of a partial last byte)
int knzo_bitstream_case(int prefix_bit, const uint8_t* arr, uint64_t nbits, uint32_t read_len, uint64_t* out) {
    KNZO_TRY
    BitWriter obs;
    if (prefix_bit >= 0) obs.writeBit(prefix_bit);
    obs.writeArray(arr, nbits);
    obs.close();
    BitReader ibs(obs.buf.data(), obs.buf.size());
    *out = ibs.readBits(read_len);
    return 0;
    KNZO_CATCH
}

// ctx["blockSize"] / ctx["entropy"] of the calling thread for the single-object and single-block entry points below (the stream entry points
// set them from their own arguments); block_size 0 / entropy 0xFFFFFFFF = key absent. Read by the TEXT transform only.
int knzo_set_ctx(uint32_t block_size, uint32_t entropy_type) { tlsBlockSize = block_size; tlsEntropyType = entropy_type; return 0; }
int knzo_get_data_type() { return tlsDataType; }
int knzo_set_data_type(int dt) { tlsDataType = dt; return 0; }

// rc -1 = transform declined (Forward error => skipped by the sequence)
int knzo_transform_forward(uint64_t t, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KNZO_TRY
    tlsDataType = DT_UNDEFINED;                       // a transform object on its own has no ctx["dataType"]
    *out_n = transformForward1(t, src, (size_t)n, dst, (size_t)cap);
    return 0;
    KNZO_CATCH
}

int knzo_transform_inverse(uint64_t t, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KNZO_TRY
    *out_n = transformInverse1(t, src, (size_t)n, dst, (size_t)cap);
    return 0;
    KNZO_CATCH
}

uint64_t knzo_max_encoded_len(uint64_t transformType, uint64_t n) {
    try { return Sequence(transformType).maxEncodedLen((size_t)n); } catch (...) { return 0; }
}

int knzo_sequence_forward(uint64_t type, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* out_n, uint8_t* skip_flags) {
    KNZO_TRY
    Sequence s(type);
    *out_n = s.forward(src, (size_t)n, dst, (size_t)cap);
    *skip_flags = s.skipFlags;
    return 0;
    KNZO_CATCH
}

int knzo_sequence_inverse(uint64_t type, uint8_t skip_flags, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KNZO_TRY
    Sequence s(type);
    s.skipFlags = skip_flags;
    *out_n = s.inverse(src, (size_t)n, dst, (size_t)cap);
    return 0;
    KNZO_CATCH
}

int knzo_encode_block(const uint8_t* src, uint64_t n, uint64_t transformType, uint32_t entropyType, int checksumBits,
                      uint8_t* out, uint64_t cap, uint64_t* out_bits, uint32_t* post_len, uint8_t* skip_flags, uint8_t* mode,
                      uint64_t* checksum) {
    KNZO_TRY
    BlockResult r;
    encodeBlock(src, (size_t)n, transformType, entropyType, checksumBits, r);
    if (r.bits.size() > cap) throw KnzError(ERR_WRITE_FILE, "output buffer too small");
    if (!r.bits.empty()) memcpy(out, r.bits.data(), r.bits.size());
    *out_bits = r.written;
    if (post_len) *post_len = (uint32_t)r.postLen;
    if (skip_flags) *skip_flags = r.skipFlags;
    if (mode) *mode = r.mode;
    if (checksum) *checksum = r.checksum;
    return 0;
    KNZO_CATCH
}

int knzo_decode_block(const uint8_t* payload, uint64_t nbytes, uint64_t transformType, uint32_t entropyType, int checksumBits,
                      uint64_t blockSize, uint8_t* out, uint64_t cap, uint64_t* out_n) {
    KNZO_TRY
    *out_n = decodeBlock(payload, (size_t)nbytes, transformType, entropyType, checksumBits, (size_t)blockSize, out, (size_t)cap);
    return 0;
    KNZO_CATCH
}

int knzo_compress(const uint8_t* src, uint64_t n, uint64_t transformType, uint32_t entropyType, uint64_t blockSize,
                  int checksumBits, int jobs, int64_t headerInputSize, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KNZO_TRY
    std::vector<uint8_t> out;
    compressStream(src, (size_t)n, transformType, entropyType, (size_t)blockSize, checksumBits, jobs, headerInputSize, out);
    *out_n = out.size();
    if (out.size() > cap) throw KnzError(ERR_WRITE_FILE, "output buffer too small");
    memcpy(dst, out.data(), out.size());
    return 0;
    KNZO_CATCH
}

// same with option flags (1 = skip incompressible blocks, the CLI's -s)
int knzo_compress2(const uint8_t* src, uint64_t n, uint64_t transformType, uint32_t entropyType, uint64_t blockSize,
                   int checksumBits, int jobs, int64_t headerInputSize, int flags, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KNZO_TRY
    std::vector<uint8_t> out;
    compressStream(src, (size_t)n, transformType, entropyType, (size_t)blockSize, checksumBits, jobs, headerInputSize, out, flags);
    *out_n = out.size();
    if (out.size() > cap) throw KnzError(ERR_WRITE_FILE, "output buffer too small");
    memcpy(dst, out.data(), out.size());
    return 0;
    KNZO_CATCH
}

uint32_t knzo_magic_type(const uint8_t* src, uint64_t n) { return getMagicType(src, (size_t)n); }
int knzo_entropy1024(const uint8_t* src, uint64_t n) {
    int histo[256] = {0};
    for (uint64_t i = 0; i < n; i++) histo[src[i]]++;
    return computeFirstOrderEntropy1024((size_t)n, histo);
}
uint32_t knzo_log2_scaled_1024(uint32_t x) { return log2ScaledBy1024(x); }

int knzo_decompress(const uint8_t* src, uint64_t n, int jobs, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KNZO_TRY
    std::vector<uint8_t> out;
    decompressStream(src, (size_t)n, jobs, out);
    *out_n = out.size();
    if (out.size() > cap) throw KnzError(ERR_WRITE_FILE, "output buffer too small");
    if (!out.empty()) memcpy(dst, out.data(), out.size());
    return 0;
    KNZO_CATCH
}

// ---- small helpers exposed for unit tests and for checking individual GPU kernels ---------------
int knzo_varint(uint32_t value, uint8_t* out8) {
    BitWriter bs;
    int n = writeVarInt(bs, value);
    bs.close();
    memcpy(out8, bs.buf.data(), bs.buf.size());
    return n;
}
uint32_t knzo_varint_read(const uint8_t* in, uint64_t nbytes, uint64_t* used_bits) {
    try { BitReader bs(in, nbytes); uint32_t v = readVarInt(bs); if (used_bits) *used_bits = bs.read(); return v; }
    catch (...) { if (used_bits) *used_bits = 0; return 0; }
}
uint32_t knzo_expgolomb_word(uint8_t val) { return val == 0 ? ((1u << 9) | 1u) : expGolombSignedWord(val); }
uint32_t knzo_xxhash32(const uint8_t* d, uint64_t n, uint32_t seed) { return xxhash32(d, (size_t)n, seed); }
uint64_t knzo_xxhash64(const uint8_t* d, uint64_t n, uint64_t seed) { return xxhash64(d, (size_t)n, seed); }

int knzo_header(int checksumBits, uint32_t entropyType, uint64_t transformType, int64_t blockSize, int64_t inputSize,
                uint8_t* out, uint64_t cap, uint64_t* out_bits) {
    KNZO_TRY
    BitWriter obs;
    writeHeader(obs, checksumBits, entropyType, transformType, blockSize, inputSize);
    *out_bits = obs.close();
    if (obs.buf.size() > cap) throw KnzError(ERR_WRITE_FILE, "output buffer too small");
    memcpy(out, obs.buf.data(), obs.buf.size());
    return 0;
    KNZO_CATCH
}

// freqs[256] in/out, returns alphabet size (negative on error)
int knzo_normalize_frequencies(int64_t* freqs, int n, int64_t total, int64_t scale, int* alphabet) {
    try { return normalizeFrequencies(freqs, n, alphabet, n, total, scale); } catch (...) { return -1; }
}

// Huffman code lengths + canonical codes for a 256-entry histogram, as updateFrequencies computes
// them (HuffmanCodec.go:128-214). codes_out[s] = (len<<12)|code. Returns the symbol count.
int knzo_huffman_codes(const int64_t* freqs_in, uint16_t* codes_out, uint8_t* header_out, uint64_t cap, uint64_t* header_bits) {
    KNZO_TRY
    int64_t f[256];
    memcpy(f, freqs_in, sizeof(f));
    BitWriter bs;
    HuffmanEncoder e(bs);
    int count = e.updateFrequencies(f);
    memcpy(codes_out, e.codes, sizeof(e.codes));
    uint64_t bits = bs.close();
    if (header_bits) *header_bits = bits;
    if (header_out) { if (bs.buf.size() > cap) throw KnzError(ERR_WRITE_FILE, "output buffer too small"); memcpy(header_out, bs.buf.data(), bs.buf.size()); }
    return count >= 0 ? 0 : -1;
    KNZO_CATCH
}

// raw BWT (no block header): dst[n], primary[8]
int knzo_bwt_forward(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t* primary8) {
    KNZO_TRY
    BWT b;
    b.forward(src, dst, (int)n);
    for (int i = 0; i < 8; i++) primary8[i] = b.primaryIndexes[i];
    return 0;
    KNZO_CATCH
}
int knzo_bwt_inverse(const uint8_t* src, uint64_t n, uint8_t* dst, const uint64_t* primary8) {
    KNZO_TRY
    BWT b;
    for (int i = 0; i < 8; i++) b.primaryIndexes[i] = primary8[i];
    b.inverse(src, dst, (int)n);
    return 0;
    KNZO_CATCH
}
int knzo_suffix_array(const uint8_t* src, uint64_t n, int32_t* sa_out) {
    KNZO_TRY
    std::vector<int32_t> sa;
    suffixArray(src, (int32_t)n, sa);
    if (n) memcpy(sa_out, sa.data(), (size_t)n * 4);
    return 0;
    KNZO_CATCH
}

// DivSufSort.go restated (oracle/divsufsort.hpp): suffix array and the BWT algorithm switch (1 = DivSufSort, 0 = SA-IS)
int knzo_suffix_array_divsufsort(const uint8_t* src, uint64_t n, int32_t* sa_out) {
    KNZO_TRY
    if (n < 2) { if (n) sa_out[0] = 0; return 0; }
    DivSufSort d;
    d.computeSuffixArray(src, sa_out, (int32_t)n);
    return 0;
    KNZO_CATCH
}
int knzo_set_bwt_algo(int algo) { bwtAlgo().store(algo ? 1 : 0); return 0; }
int knzo_get_bwt_algo() { return bwtAlgo().load(); }

} // extern "C"
