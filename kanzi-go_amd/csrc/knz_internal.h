// Host-side internals of libknz_gpu: the GPU batch scheduler that Writer.processBlock / Reader.processBlock are
// re-pointed at (v2/io/CompressedStream.go:621-710, 1614-1744). One Handle = one device workspace + one HIP stream.
#pragma once
#include "../../include/knz_gpu.h"
#include "bits.h"
#include <string>
#include <vector>

// The library reads the environment in exactly two places. knz_test_switch: forms the test suites select (exact fall-backs, A/B of the
// hand-written loops, small geometry on small inputs). knz_measure_switch: measurement variants and diagnostics, compiled in only with
// -DKNZ_MEASURE (the shipped library carries the default form of every kernel plus the fall-backs the tests need, nothing else).
#include <cstdlib>
static inline const char* knz_test_switch(const char* name) { return getenv(name); }
#ifdef KNZ_MEASURE
static inline const char* knz_measure_switch(const char* name) { return getenv(name); }
#else
static inline const char* knz_measure_switch(const char*) { return nullptr; }
#endif

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    DevBuf();                         // (registers itself with the Handle under construction: knz_release_workspace walks the list)
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }          // every workspace buffer of a Handle goes with it (knz_close)
    int reserve(size_t n);            // grows (never shrinks); contents are not preserved
    void release();
    template <typename T> T* as() const { return (T*)p; }
};

// HIP-event pair around one launch of a kernel that can dominate a batch (bench.py's roofline line reads these)
#define KNZ_MAX_PROBES 128
struct KernelProbe { const char* name = nullptr; hipEvent_t a = nullptr, b = nullptr; };

struct MultiDev;                      // several lanes (devices, or streams of one device) behind one handle: knz_multi.inc

struct Handle {
    knz_cfg cfg;
    MultiDev* multi = nullptr;        // != nullptr: a handle of knz_open_devices(); the batch calls fan out over its lanes, the workspace below stays empty
    KernelProbe probes[KNZ_MAX_PROBES];
    int nprobes = 0;
    uint64_t stage_bytes[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // bytes that entered transform stage i of the last encode batch (knz_last_counter 8 + i)
    DevBuf stage_sum;                 // their device-side sums
    uint64_t post_bytes = 0;          // bytes behind the transform sequence of the last encode batch (entropy coder input)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t hstream = nullptr;    // the stream of the host-pointer entry points: non-blocking (nothing of the caller's lives on a device stream there)
    std::string err;
    // workspace
    DevBuf blk_off, blk_len, blk_src_len, blk_skip, blk_cksum, blk_status;
    DevBuf unit_bits, unit_src, ans_tab, scratch, chunk_rel, blk_written, blk_hdr, blk_dst_bit, total_bits;
    DevBuf stage_in, stage_out;       // host-pointer entry points stage through these
    DevBuf dec_tables;                // decoder per-chunk positions
    DevBuf utf_map, utf_syms, utf_ranks, utf_inv, utf_bits, blk_dt;   // UTF codec: alias table / code points / ranks / inverse map / walk bitmap, ctx["dataType"] per block
    DevBuf text_tok, text_pos, text_instok, text_cnt, text_prof;                           // parallel TEXT: per-token arrays, per-position marks, entry -> token
    DevBuf text_map, text_ent, text_stat, text_mode;                  // TEXT codec: hash tables, dictionary entries, static dictionary (uploaded once), mode bytes
    bool text_stat_ready = false;
    DevBuf srt_tab, srt_tmp, srt_ptrs;                               // parallel SRT forward: per-block tables, MTFT ranks, pointer / dummy arrays
    DevBuf blk_copy;                                                 // [nblocks] 1 = copy block (<= 15 bytes, or skipped by -s)
    DevBuf huf_fhist;                                                // symbol counts per fragment (sizes pass of the Huffman encoder)
    DevBuf huf_stfreq, huf_stsym, huf_stlen, huf_stcnt, huf_stmax;   // sorted chunk statistics between the Huffman encode kernels
    size_t huf_fallback_n = 0;        // chunks covered by huf_fallback in the last decode batch
    DevBuf huf_fallback;              // [chunks] 1 = the parallel Huffman decoder handed the chunk to the serial one
    // transform pipeline (knz_transforms.inc)
    DevBuf xf_r3, pipe_prog, pipe_flag, pipe_group;                              // fused ZRLT / RANK inverse under the rANS-1 decoder (rank_pipe.hip): rank region, progress words, per-block done flags
    hipStream_t stream2 = nullptr;                                  // its stream (non-blocking) and the two events that tie it to the caller's
    bool pipe_ready = false;
    size_t pipe_n = 0;                // blocks covered by pipe_flag in the last decode batch (0: the fused path was not taken)
    hipStream_t stream3 = nullptr;                                  // ... and the stream of the launch that holds the long chains
    hipEvent_t ev_pipe[3] = {nullptr, nullptr, nullptr};
    DevBuf xf_r1, xf_r2, xf_outptr, xf_outlen, xf_ok, xf_side, xf_active, xf_take, xf_sega, xf_segb, xf_gstart, xf_misc;
    DevBuf lz_hash, lz_tk, lz_mb, lz_ml;
    DevBuf lz_k0, lz_k1, lz_v0, lz_v1, lz_cand, lz_cp, lz_holes, lz_gstart;     // second form of the LZ forward (lz_par.hip): sort buffers, candidates, common prefixes, hole bitmaps
    DevBuf lzi_geo, lzi_tokbase, lzi_ta, lzi_tb, lzi_tc, lzi_td, lzi_seg, lzi_lxg, lzi_lxc, lzi_ml, lzi_map, lzi_flag, lzi_serial;   // parallel LZ inverse (lz_inv_par.hip)
    DevBuf lzs_state, lzs_tok, lzs_maps, lzs_misc, lzs_mask, lzs_q, lzs_chg;     // segment-parallel LZ forward (lz_fwd_seg.hip): entry / exit states, token descriptors, hole maps, per-block tables
    size_t lzs_n = 0;                 // blocks covered by lzs_misc's state bytes in the last LZ forward stage
    uint32_t lzs_rounds = 0;          // rounds its fixed point took
    size_t lzi_serial_n = 0;          // blocks covered by lzi_serial in the last LZ inverse stage (1 = went to the one-wave kernel)
    DevBuf a1_freqs, a1_tab, a1_ctxhdr, a1_ctxbits, a1_dtab, a1_info, a1_paybit, a1_f16, a1_ent, a1_cum, a1_ctxpos;
    DevBuf sa_keys0, sa_keys1, sa_vals0, sa_vals1, sa_rank, sa_gs, sa_head, sa_unres, sa_pos, sa_tmp, sa_links, sa_sp;
    DevBuf sa_hb, sa_tiles, sa_posl0, sa_posl1, sa_gid0, sa_gid1;   // suffix sort (bwt_sort.hip): head bits, per-tile tables, the large list
    void* pinned = nullptr;           // small pinned host area for results
    // results of the last encode batch, one row per block + the batch totals behind them: packed on the device (knz_pack_results_kernel), ONE copy into pinned memory
    struct ResultRow { uint64_t written; uint64_t cksum; uint32_t post_len; int32_t status; uint32_t mode; uint32_t skip; };
    DevBuf res_rows;
    ResultRow* pinned_rows = nullptr; // grow-only
    size_t pinned_rows_cap = 0;
    int reserve_pinned_rows(size_t n) {
        if (n + 1 <= pinned_rows_cap) return 0;
        if (pinned_rows) hipHostFree(pinned_rows);
        pinned_rows = nullptr; pinned_rows_cap = 0;
        const size_t want = n + n / 4 + 64;
        if (hipHostMalloc((void**)&pinned_rows, sizeof(ResultRow) * want) != hipSuccess) return -1;
        pinned_rows_cap = want;
        return 0;
    }
    std::vector<DevBuf*> all_bufs;    // every workspace buffer of this handle (filled while the handle is constructed)
    hipEvent_t ev[KNZ_STAGE_COUNT + 1];
    bool ev_valid = false;
    float stage_ms[KNZ_STAGE_COUNT];
};

// ---- kernels (defined in the .hip files) -------------------------------------------------------------------------
struct HufEncArgs;
struct LayoutArgs;
struct StreamArgs;
struct GatherArgs;

int knz_set_error(Handle* h, int code, const char* msg);

// stream header (v2/io/CompressedStream.go:429-519): returns bit count, fills BE words
uint32_t knz_build_stream_header(const knz_cfg& cfg, int64_t inputSize, uint32_t words[8]);
