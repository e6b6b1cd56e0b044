/*
 * knz_gpu.h — C ABI of the MI355X-native Kanzi block-compression hot path.
 *
 * Drop-in boundary for flanglet/kanzi-go (bitstream v6). Bit-exactness is pinned by a MECHANICAL TRANSLATION of the reference, not yet by a Go-built
 * binary: the image has no Go toolchain, so kanzi-go's .go sources are translated to C++ by tools/go2cpp and compiled (oracle/_ref); the device is compared
 * with that build directly, through 387 stream vectors it wrote (tests/test_ref_streams.py), through the full-size BASELINE streams by sha256
 * (tests/golden/ref_streams/fullsize_manifest.json), and with the hand-written restatement (oracle/) that _ref pins case by case. A misreading of Go shared by
 * the translator and the restatement would pass all of that; a stream from a Go-compiled build is the missing witness (tools/make_ref_vectors.sh writes the
 * same manifests on any machine with Go). Every entry point names the reference interface it replaces; the cgo stubs a kanzi-go maintainer would add are
 * in INTEGRATION.md and go/.
 * Plain pointers and sizes only; the library never keeps a caller pointer after a call returns.
 * Return value: 0 on success, otherwise a kanzi error code (v2/Definitions.go:25-46), except
 * knz_transform_forward() which returns KNZ_SKIP when the transform declines (in kanzi-go a
 * Forward error means "skip this transform", v2/transform/Sequence.go:86-91).
 *
 * The product library (libknz_gpu.so) contains gfx950 code only. There is no CPU fallback: with no
 * usable GPU knz_open() fails with KNZ_ERR_CREATE_COMPRESSOR.
 */
#ifndef KNZ_GPU_H
#define KNZ_GPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* v2/Definitions.go:25-46 */
enum {
    KNZ_OK = 0,
    KNZ_ERR_MISSING_PARAM = 1, KNZ_ERR_BLOCK_SIZE = 2, KNZ_ERR_INVALID_CODEC = 3,
    KNZ_ERR_CREATE_COMPRESSOR = 4, KNZ_ERR_CREATE_DECOMPRESSOR = 5, KNZ_ERR_READ_FILE = 11,
    KNZ_ERR_WRITE_FILE = 12, KNZ_ERR_PROCESS_BLOCK = 13, KNZ_ERR_CREATE_CODEC = 14,
    KNZ_ERR_INVALID_FILE = 15, KNZ_ERR_STREAM_VERSION = 16, KNZ_ERR_INVALID_PARAM = 18,
    KNZ_ERR_CRC_CHECK = 19, KNZ_ERR_UNKNOWN = 127,
    KNZ_SKIP = -1
};

/* knz_cfg.flags: KNZ_FLAG_SKIP_BLOCKS = the CLI's -s / ctx["skipBlocks"] (v2/io/CompressedStream.go:778-800): blocks that start
 * with the magic number of a compressed format or whose order-0 entropy is >= 973/1024 are emitted as copy blocks */
enum { KNZ_FLAG_SKIP_BLOCKS = 1 };

/* transform ids, v2/transform/Factory.go:31-53 (6 bits each, first transform in bits 47..42). KNZ_T_TEXT is DICT_TYPE ("TEXT"): which of
 * its two stream formats is written / read follows knz_cfg.entropy (Factory.go:100-120: NONE, HUFFMAN, RANGE, ANS0 -> the byte-oriented one),
 * its hash size follows knz_cfg.block_size (TextCodec.go:610-650, :1137-1188), as ctx["entropy"] / ctx["blockSize"] do in the reference; the data
 * type it detects (ctx["dataType"]) reaches the UTF and LZ stages behind it in the same sequence. */
enum { KNZ_T_NONE = 0, KNZ_T_BWT = 1, KNZ_T_LZ = 3, KNZ_T_ZRLT = 6, KNZ_T_MTFT = 7, KNZ_T_RANK = 8, KNZ_T_TEXT = 10, KNZ_T_SRT = 13, KNZ_T_LZP = 14, KNZ_T_LZX = 16, KNZ_T_UTF = 17 };
/* entropy ids, v2/entropy/EntropyCodecFactory.go:26-42 */
enum { KNZ_E_NONE = 0, KNZ_E_HUFFMAN = 1, KNZ_E_FPAQ = 2, KNZ_E_ANS0 = 5, KNZ_E_ANS1 = 8 };

/* Flat form of the ctx map handed to every kanzi-go factory (v2/io/CompressedStream.go:217-224,370-382). */
typedef struct {
    uint64_t transform;      /* packed as transform.GetType() returns it (Factory.go:289-328) */
    uint32_t entropy;        /* entropy.GetType() value                                      */
    uint32_t block_size;     /* ctx["blockSize"], multiple of 16, 1 KiB .. 1 GiB              */
    uint32_t checksum_bits;  /* 0, 32 or 64 (Writer hasher32/hasher64)                        */
    uint32_t bs_version;     /* ctx["bsVersion"]; only 6 is produced/accepted                 */
    int32_t  device;         /* HIP device ordinal, -1 = current device                       */
    uint32_t flags;          /* KNZ_FLAG_* bits                                               */
} knz_cfg;

/* One block of a batch: what one encodingTask/decodingTask owns (CompressedStream.go:189-214,1020-1045). */
typedef struct {
    const uint8_t* src;      /* encode: block bytes ; decode: block-local payload (mode byte first) */
    uint32_t src_len;        /* bytes                                                          */
    uint8_t* dst;            /* encode: block-local stream ; decode: decoded bytes              */
    uint32_t dst_cap;
    uint64_t out_bits;       /* encode: exact bit count (obs.Written() after Close, :912-914) ; decode: decoded bytes */
    uint32_t post_len;       /* post-transform length (EVT_AFTER_TRANSFORM size)               */
    uint8_t  skip_flags;     /* ByteTransformSequence.SkipFlags()                              */
    uint8_t  mode;           /* block mode byte                                                */
    uint16_t reserved;
    uint64_t checksum;       /* XXHash32/64 of the original block when checksum_bits != 0       */
    int32_t  status;         /* per-block kanzi error code (0 = ok)                            */
    int32_t  reserved2;
} knz_block;

/* Handle = one GPU batch scheduler (device workspace + stream). One per io.Writer / io.Reader. */
int knz_open(const knz_cfg* cfg, void** handle);
int knz_close(void* handle);
const char* knz_last_error(void* handle);

/*
 * One handle over several devices: the `jobs`-wide goroutine fan-out of Writer.processBlock / Reader.processBlock
 * (v2/io/CompressedStream.go:621-710, :1614-1744) becomes a fan-out over GPUs inside knz_encode_blocks / knz_decode_blocks (SURVEY 8b's
 * device_mask, as a list). ordinals[0..n): HIP device ordinals, one LANE each (own workspace, own stream, own host worker thread), 1 <= n <= 64.
 * An ordinal may be named several times: its lanes share the device, and the uploads / downloads of one run under the kernels of another.
 * A batch is cut into n contiguous balanced ranges (the first nblocks % n lanes take one block more: 26 blocks over 8 lanes = 4,4,3,3,3,3,3,3);
 * every lane copies its blocks in, runs them and copies the results straight into the caller's dst. Blocks are independent
 * (v2/Definitions.go:73-77, io/CompressedStream.go:896-898): no collective is involved and the bytes are those of a one-device handle.
 * Per-block status and the first failing block's error code are reported as by a one-device batch. The other entry points accept such a handle too:
 * single-object calls run on the first lane, device-resident calls (knz_dev_*) on the lane whose device owns d_dst.
 * cfg->device is ignored. knz_device_count() = hipGetDeviceCount (0 without a usable GPU).
 */
int knz_open_devices(const knz_cfg* cfg, const int32_t* ordinals, int n, void** handle);
int knz_device_count(void);
/* lanes behind a handle (1 for a knz_open handle) ; per lane of the last batch call: device ordinal, blocks taken, wall-clock ms (upload .. download). Returns lanes filled. */
int knz_lane_count(void* handle);
int knz_last_lane_times(void* handle, int32_t* devices, int32_t* blocks, float* ms, int cap);

/*
 * Replaces the goroutine fan-out of Writer.processBlock (v2/io/CompressedStream.go:636-701): the n
 * buffered blocks are encoded in one device batch. For every block the result equals
 * encodingTask.encode up to obs.Close() (:729-914): dst holds mode byte .. entropy payload, out_bits the
 * exact bit count. The Go host then performs the ordered emission (:951-976).
 */
int knz_encode_blocks(void* handle, knz_block* blocks, int n);

/*
 * Replaces the concurrent part of Reader.processBlock / decodingTask.decode after the payload has been
 * read from the shared stream (:1875-2011). src = payload bytes (r = (read+7)>>3 of them).
 */
int knz_decode_blocks(void* handle, knz_block* blocks, int n);

/*
 * Whole-stream, device-resident form used by the bench and by multi-GPU sharding: d_src/d_dst are
 * DEVICE pointers. Produces/consumes the complete .knz stream (header :429-519, blocks, end marker
 * :593-594). hip_stream is a hipStream_t (NULL = the handle's own stream). out_bytes is a host pointer.
 * header_input_size is ctx["fileSize"] written to the stream header (0 = unknown).
 * Compressed streams are read as big-endian 32-bit words: d_src of the decode calls (and d_dst of the encode calls) must be 4-byte aligned
 * and readable / writable up to the next multiple of 4 bytes behind n_bytes / dst_cap.
 */
int knz_dev_compress(void* handle, const void* d_src, uint64_t n, int64_t header_input_size,
                     void* d_dst, uint64_t dst_cap, uint64_t* out_bytes, void* hip_stream);
int knz_dev_decompress(void* handle, const void* d_src, uint64_t n_bytes,
                       void* d_dst, uint64_t dst_cap, uint64_t* out_bytes, void* hip_stream);

/*
 * Multi-GPU block sharding: a rank encodes blocks [first_block, first_block+n_blocks) of a
 * stream whose blocks are block_size bytes each; d_src points at this rank's first block. The result is a
 * bit string (no stream header, no end marker): *out_bits bits, zero padded to a byte in d_dst.
 * knz_dev_assemble() then concatenates the ranks' bit strings, in rank order, behind the stream header
 * and appends the end marker (it is the device form of the ordered emission, :934-976).
 */
int knz_dev_compress_blocks(void* handle, const void* d_src, uint64_t n, void* d_dst, uint64_t dst_cap,
                            uint64_t* out_bits, void* hip_stream);
/* Decode side of the sharding: a rank decodes the framed blocks of its own segment (n_bits bits, no header). */
int knz_dev_decompress_blocks(void* handle, const void* d_src, uint64_t n_bits, void* d_dst, uint64_t dst_cap,
                              uint64_t* out_bytes, void* hip_stream);
int knz_dev_assemble(void* handle, int64_t header_input_size, const void* const* d_segments,
                     const uint64_t* segment_bits, int n_segments, void* d_dst, uint64_t dst_cap,
                     uint64_t* out_bytes, void* hip_stream);

/* Single kanzi.ByteTransform objects (v2/Definitions.go:78-91). type1 = one 6-bit transform id. Host pointers. */
int knz_transform_forward(void* handle, uint64_t type1, const uint8_t* src, uint32_t n,
                          uint8_t* dst, uint32_t cap, uint32_t* out_n);
int knz_transform_inverse(void* handle, uint64_t type1, const uint8_t* src, uint32_t n,
                          uint8_t* dst, uint32_t cap, uint32_t* out_n);
/* ByteTransform.MaxEncodedLen for a packed sequence (Sequence.go:189-205) */
uint32_t knz_max_encoded_len(uint64_t transform, uint32_t n);

/* Single kanzi.EntropyEncoder / EntropyDecoder objects (v2/Definitions.go:154-179). Host pointers.
 * encode: the shim then calls obs.WriteArray(bits, out_bits). decode: bits = the remaining payload. */
int knz_entropy_encode(void* handle, uint32_t type, const uint8_t* src, uint32_t n,
                       uint8_t* bits, uint64_t cap_bytes, uint64_t* out_bits);
int knz_entropy_decode(void* handle, uint32_t type, const uint8_t* bits, uint64_t n_bytes,
                       uint8_t* dst, uint32_t n, uint64_t* used_bits);

/* Timing of the last device batch, measured with HIP events on the stream the kernels ran on.
 * kernel_ms[] receives per-stage times, see KNZ_STAGE_*; returns the number of stages filled. */
enum { KNZ_STAGE_TRANSFORM = 0, KNZ_STAGE_ENTROPY = 1, KNZ_STAGE_LAYOUT = 2, KNZ_STAGE_GATHER = 3, KNZ_STAGE_COUNT = 4 };
int knz_last_timing(void* handle, float* stage_ms, int cap);

/* Per-kernel times of the last device batch: the launches that can dominate a batch (the per-block chains, the entropy
 * kernels) are bracketed by a HIP event pair on the launch stream. names receives the kernel names separated by '\n'
 * (launch order, a kernel launched several times appears several times), ms[i] the duration of launch i. Returns the
 * number of launches reported. bench.py's roofline line is built from these. */
int knz_last_kernel_times(void* handle, char* names, int names_cap, float* ms, int cap);

/* Diagnostic counters of the last device batch. KNZ_COUNTER_HUF_SERIAL_CHUNKS: Huffman chunks the wave-parallel decoder
 * handed back to the serial (reference-order) decoder; 0 for any stream a kanzi encoder wrote. Returns 0 or an error code. */
enum { KNZ_COUNTER_HUF_SERIAL_CHUNKS = 0, KNZ_COUNTER_POST_TRANSFORM_BYTES = 1 /* entropy coder input of the last encode batch */,
       KNZ_COUNTER_TEXT_CHAIN_BLOCKS = 2 /* blocks of the last TEXT stage scanned by the one-lane kernel instead of the parallel one */,
       KNZ_COUNTER_LZ_INV_SERIAL_BLOCKS = 3 /* blocks of the last LZ / LZX inverse stage decoded by the one-wave kernel instead of the parallel one */,
       KNZ_COUNTER_LZ_FWD_SERIAL_BLOCKS = 4 /* blocks of the last LZ / LZX forward stage whose segment-parallel parse did not settle (or that need a
                                               decision the parallel layout leaves to the one-wave kernel): parsed by the one-wave kernel */,
       KNZ_COUNTER_LZ_FWD_ROUNDS = 5 /* rounds the fixed point of the last segment-parallel LZ forward stage took */,
       KNZ_COUNTER_RANK_PIPE_BLOCKS = 6 /* blocks of the last decode batch whose ZRLT / RANK inverses ran as one chain under the order-1 rANS decoder */,
       KNZ_COUNTER_STAGE_BYTES0 = 8 /* 8 + i: bytes that entered transform stage i (0..7) of the last encode batch, summed over its blocks */ };
int knz_last_counter(void* handle, int id, uint64_t* value);

/* 1 when a transform/entropy id has a device implementation in this build */
int knz_supports(uint64_t transform, uint32_t entropy);

#ifdef __cplusplus
}
#endif
#endif /* KNZ_GPU_H */
