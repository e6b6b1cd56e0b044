// libknz_gpu: C ABI (include/knz_gpu.h) + GPU batch scheduler. Unity build of the gfx950 kernels.
// Product code: gfx950 only, no CPU fallback (knz_open fails without a GPU).
#include "knz_internal.h"
#include "huffman_enc.hip"
#include "huffman_dec.hip"
#include "huffman_par.hip"
#include "ans0.hip"
#include "ans1.hip"
#include "fpaq.hip"
#include "transforms.hip"
#include "rank_inv.hip"
#include "rank_pipe.hip"
#include "bwt.hip"
#include "lz.hip"
#include "lz_par.hip"
#include "lz_inv_par.hip"
#include "lz_fwd_seg.hip"
#include "srt_lzp.hip"
#include "text.hip"
#include "text_par.hip"
#include "utf.hip"
#include "xxhash.hip"
#include "skip.hip"
#include "prims.h"
#include "bwt_sort.hip"
#include "layout.hip"
#include <algorithm>
#include <cstdio>
#include <cstring>

#define HIP_OK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) return knz_set_error(h, KNZ_ERR_UNKNOWN, hipGetErrorString(e__)); \
    } while (0)

static int probe_begin(Handle* h, hipStream_t st, const char* name) {
    if (h->nprobes >= KNZ_MAX_PROBES) return -1;
    KernelProbe& p = h->probes[h->nprobes];
    if (!p.a && (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess)) return -1;
    p.name = name;
    hipEventRecord(p.a, st);
    return h->nprobes++;
}
static void probe_end(Handle* h, hipStream_t st, int id) { if (id >= 0) hipEventRecord(h->probes[id].b, st); }
// launch with a probe: `h` and `st` are in scope at every call site
#define KNZ_LAUNCH_PROBED(kern, ...)                         \
    do {                                                      \
        const int pid__ = probe_begin(h, st, #kern);          \
        hipLaunchKernelGGL(kern, __VA_ARGS__);                \
        probe_end(h, st, pid__);                              \
    } while (0)

// HIP's current device is a per-host-thread setting: every entry point binds the calling thread to the handle's device for the
// duration of the call (a handle opened on device N may be used from any goroutine / OS thread) and restores what was there.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(const Handle* h) {
        if (h && hipGetDevice(&prev) == hipSuccess && prev != h->device) switched = hipSetDevice(h->device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// several lanes behind one handle (knz_multi.inc)
static int multi_blocks(Handle* h, knz_block* blocks, int n, int job);
static void multi_close(Handle* h);
static Handle* lane0(Handle* h);
static Handle* lane_of_pointer(Handle* h, const void* d_ptr);

int knz_set_error(Handle* h, int code, const char* msg) {
    if (h) h->err = msg ? msg : "";
    return code;
}

// A workspace growth of 64 MiB or more is refused while it would leave the device with less than 1/16 of its memory (4 GiB at least): a device driven to its
// last byte takes the HIP runtime down with it (queue creation aborts the process), and several handles share one device. A refused
// growth comes back as KNZ_ERR_CREATE_COMPRESSOR / _DECOMPRESSOR; the host-pointer entry points then release the workspace and take the
// batch in halves (knz_host_api.inc). KNZ_TEST_ALLOC_LIMIT (tests): bytes one buffer may hold.
static thread_local std::vector<DevBuf*>* g_buf_registry = nullptr;
static thread_local bool g_alloc_refused = false;
DevBuf::DevBuf() { if (g_buf_registry) g_buf_registry->push_back(this); }
int DevBuf::reserve(size_t n) {
    if (n <= cap) return 0;
    // (every check comes before the old buffer is given back: a refused growth leaves the handle as it was)
    const size_t want = n + n / 8 + 256;
    if (const char* lim = knz_test_switch("KNZ_TEST_ALLOC_LIMIT")) { if (want > (size_t)strtoull(lim, nullptr, 10)) { g_alloc_refused = true; return -1; } }
    size_t freeB = 0, totalB = 0;
    if (want - cap >= ((size_t)64 << 20) && hipMemGetInfo(&freeB, &totalB) == hipSuccess && totalB != 0) {   // the headroom rule is for growths that matter (>= 64 MiB)
        const size_t keep = std::min(std::max<size_t>(totalB / 16, (size_t)4 << 30), totalB / 2);
        const size_t avail = freeB + cap;                 // (the old buffer goes first)
        if (want > avail || avail - want < keep) { g_alloc_refused = true; return -1; }
    }
    if (p) hipFree(p);
    p = nullptr; cap = 0;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; (void)hipGetLastError(); g_alloc_refused = true; return -1; }
    cap = want;
    return 0;
}
void DevBuf::release() { if (p) hipFree(p); p = nullptr; cap = 0; }
static void knz_release_workspace(Handle* h) {
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->hstream) hipStreamSynchronize(h->hstream);
    if (h->pipe_ready) { hipStreamSynchronize(h->stream2); hipStreamSynchronize(h->stream3); }
    for (DevBuf* b : h->all_bufs) b->release();
    h->text_stat_ready = false;                      // (the static TEXT dictionary lives in one of them: uploaded again on demand)
    h->huf_fallback_n = 0; h->lzs_n = 0; h->lzi_serial_n = 0; h->pipe_n = 0;
}

// The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and kernels of streams that share
// a queue run one after the other. A Go host keeps one handle per io.Writer / io.Reader (two streams each: the caller's and the fused ZRLT / RANK
// chain's), so four queues serialise everything beyond two handles (measured: profiles/r05_multi_handle.json). The variable is read when the
// runtime initialises: set here, when the library is loaded, unless the host has chosen a value (no effect if HIP is already up).
#ifndef KNZ_HIP_EMU
__attribute__((constructor)) static void knz_more_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "32", 0); }
#endif

// ---- stream header, v2/io/CompressedStream.go:429-519 -----------------------------------------------------------
static void hdr_put(uint32_t* words, uint32_t& pos, uint64_t value, uint32_t count) {
    for (int i = (int)count - 1; i >= 0; i--) {      // host side, a couple of hundred bits: bit by bit is fine
        uint32_t bit = (uint32_t)((value >> i) & 1);
        words[pos >> 5] |= bit << (31 - (pos & 31));
        pos++;
    }
}
uint32_t knz_build_stream_header(const knz_cfg& cfg, int64_t inputSize, uint32_t words[8]) {
    for (int i = 0; i < 8; i++) words[i] = 0;
    uint32_t pos = 0;
    const int ckSize = cfg.checksum_bits == 32 ? 1 : (cfg.checksum_bits == 64 ? 2 : 0);
    hdr_put(words, pos, 0x4B414E5Au, 32);              // _BITSTREAM_TYPE
    hdr_put(words, pos, 6, 4);                         // _BITSTREAM_FORMAT_VERSION
    hdr_put(words, pos, (uint64_t)ckSize, 2);
    hdr_put(words, pos, cfg.entropy, 5);
    hdr_put(words, pos, cfg.transform, 48);
    hdr_put(words, pos, cfg.block_size >> 4, 28);
    uint32_t szMask;
    if (inputSize <= 0 || inputSize >= ((int64_t)1 << 48)) szMask = 0;
    else if (inputSize >= ((int64_t)1 << 32)) szMask = 3;
    else if (inputSize >= ((int64_t)1 << 16)) szMask = 2;
    else szMask = 1;
    hdr_put(words, pos, szMask, 2);
    if (szMask) hdr_put(words, pos, (uint64_t)inputSize, 16 * szMask);
    hdr_put(words, pos, 0, 15);
    const uint32_t HASH = 0x1E35A7BDu;
    uint32_t ck = HASH * (0x01030507u * 6u);
    ck ^= HASH * (uint32_t)(~(uint32_t)ckSize);
    ck ^= HASH * (uint32_t)(~cfg.entropy);
    ck ^= HASH * (uint32_t)((~cfg.transform) >> 32);
    ck ^= HASH * (uint32_t)(~cfg.transform);
    ck ^= HASH * (uint32_t)(~cfg.block_size);
    if (szMask) {
        ck ^= HASH * (uint32_t)((~(uint64_t)inputSize) >> 32);
        ck ^= HASH * (uint32_t)(~(uint64_t)inputSize);
    }
    ck = (ck >> 23) ^ (ck >> 3);
    hdr_put(words, pos, ck & 0xFFFFFF, 24);
    return pos;
}

// ---- capability table ----------------------------------------------------------------------------------------------
static bool transform_on_device(uint64_t t) {                    // packed sequence
    for (int s = 42; s >= 0; s -= 6) {
        const uint32_t id = (uint32_t)((t >> s) & 63);
        if (!(id == KNZ_T_NONE || id == KNZ_T_BWT || id == KNZ_T_RANK || id == KNZ_T_MTFT || id == KNZ_T_ZRLT || id == KNZ_T_LZ || id == KNZ_T_LZX || id == KNZ_T_SRT || id == KNZ_T_LZP || id == KNZ_T_UTF || id == KNZ_T_TEXT)) return false;
    }
    return true;
}
static bool entropy_on_device(uint32_t e) { return e == KNZ_E_HUFFMAN || e == KNZ_E_NONE || e == KNZ_E_ANS0 || e == KNZ_E_ANS1 || e == KNZ_E_FPAQ; }

extern "C" int knz_supports(uint64_t transform, uint32_t entropy) {
    return (transform_on_device(transform) && entropy_on_device(entropy)) ? 1 : 0;
}

static uint32_t seq_len(uint64_t t) {
    uint32_t n = 0;
    for (int s = 42; s >= 0; s -= 6) if ((t >> s) & 63) n++;
    return n ? n : 1;
}

extern "C" uint32_t knz_max_encoded_len(uint64_t transform, uint32_t n) {
    // Sequence.go:189-205 over the hot-path transforms (BWT/SBRT: n+33, LZ/LZX/LZP: n+16 | n+n/64, SRT: n+1024)
    uint64_t req = n;
    for (int s = 42; s >= 0; s -= 6) {
        uint32_t t = (uint32_t)((transform >> s) & 63);
        uint64_t nxt = req;
        if (t == KNZ_T_BWT || t == KNZ_T_RANK || t == KNZ_T_MTFT) nxt = req + 33;
        else if (t == KNZ_T_LZ || t == KNZ_T_LZX || t == KNZ_T_LZP) nxt = req <= 1024 ? req + 16 : req + req / 64;
        else if (t == KNZ_T_SRT) nxt = req + 4 * 256;
        else if (t == KNZ_T_UTF) nxt = req + 8192;
        if (nxt > req) req = nxt;
    }
    return (uint32_t)std::min<uint64_t>(req, 0xFFFFFFFFu);
}

// ---- open / close --------------------------------------------------------------------------------------------------
static thread_local std::string g_open_error;

extern "C" int knz_open(const knz_cfg* cfg, void** handle) {
    if (!cfg || !handle) return KNZ_ERR_MISSING_PARAM;
    *handle = nullptr;
    g_open_error.clear();
    if (cfg->block_size < 1024 || cfg->block_size > (1u << 30) || (cfg->block_size & 15)) { g_open_error = "invalid block size"; return KNZ_ERR_BLOCK_SIZE; }
    if (cfg->checksum_bits != 0 && cfg->checksum_bits != 32 && cfg->checksum_bits != 64) { g_open_error = "invalid checksum size"; return KNZ_ERR_INVALID_PARAM; }
    if (cfg->bs_version != 0 && cfg->bs_version != 6) { g_open_error = "only bitstream version 6"; return KNZ_ERR_STREAM_VERSION; }
    std::vector<DevBuf*> bufs;
    g_buf_registry = &bufs;
    Handle* h = new Handle();
    g_buf_registry = nullptr;
    h->all_bufs = bufs;
    h->cfg = *cfg;
    h->cfg.bs_version = 6;
    int dev = cfg->device;
    hipError_t e = hipSuccess;
    if (dev < 0) { e = hipGetDevice(&dev); if (e != hipSuccess) dev = 0; }
    e = hipSetDevice(dev);
    if (e != hipSuccess) { g_open_error = std::string("hipSetDevice: ") + hipGetErrorString(e); delete h; return KNZ_ERR_CREATE_COMPRESSOR; }
    h->device = dev;
    void* probe = nullptr;
    e = hipMalloc(&probe, 256);                      // no usable GPU: fail loudly, there is no CPU fallback
    if (e != hipSuccess) { g_open_error = std::string("hipMalloc: ") + hipGetErrorString(e); delete h; return KNZ_ERR_CREATE_COMPRESSOR; }
    hipFree(probe);
    e = hipHostMalloc(&h->pinned, 4096);
    if (e != hipSuccess) { g_open_error = std::string("hipHostMalloc: ") + hipGetErrorString(e); delete h; return KNZ_ERR_CREATE_COMPRESSOR; }
    // the handle's own stream: a blocking stream, i.e. ordered against the null stream (callers that prepare buffers on the
    // default stream, PyTorch included, need no extra synchronisation) but not against the streams of other handles
    if (hipStreamCreateWithFlags(&h->stream, hipStreamDefault) == hipSuccess) h->own_stream = true;
    else h->stream = nullptr;
    // ... and one for the host-pointer entry points (knz_encode_blocks, knz_decode_blocks, the single-object calls): every buffer of the caller is host
    // memory there, so nothing has to be ordered against the NULL stream of whatever else lives in the process
    if (hipStreamCreateWithFlags(&h->hstream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); h->hstream = nullptr; }
    for (int i = 0; i <= KNZ_STAGE_COUNT; i++) hipEventCreate(&h->ev[i]);
    // The two streams of the fused ZRLT / RANK chain (rank_pipe.hip) get priorities of their own: the runtime keeps a pool of hardware queues per priority
    // level, so the long chains (stream3, highest), the short ones (stream2, lowest) and the caller's stream (the decoder under which they start) can never
    // share a hardware queue. At one priority the runtime deals its few queues out in turn to every stream the PROCESS creates: where two of the three landed
    // on one queue the kernels ran one after the other (a decode of configs[3] in 833 ms instead of 525, seen with the handle's own stream in a torch process).
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); least = greatest = 0; }
        const bool s2 = (least != greatest && hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, least) == hipSuccess) ||
                        ((void)hipGetLastError(), hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking) == hipSuccess);
        const bool s3 = (least != greatest && hipStreamCreateWithPriority(&h->stream3, hipStreamNonBlocking, greatest) == hipSuccess) ||
                        ((void)hipGetLastError(), hipStreamCreateWithFlags(&h->stream3, hipStreamNonBlocking) == hipSuccess);
        h->pipe_ready = s2 && s3;
    }
    for (int i = 0; i < 3; i++) if (hipEventCreateWithFlags(&h->ev_pipe[i], hipEventDisableTiming) != hipSuccess) h->pipe_ready = false;
    for (int i = 0; i < KNZ_STAGE_COUNT; i++) h->stage_ms[i] = 0.f;
    *handle = h;
    return KNZ_OK;
}

extern "C" int knz_close(void* handle) {
    Handle* h = (Handle*)handle;
    if (!h) return KNZ_OK;
    if (h->multi) { multi_close(h); delete h; return KNZ_OK; }
    DeviceGuard dg(h);                                  // (the workspace buffers are freed by ~Handle while the device is bound)
    if (h->own_stream) { hipStreamSynchronize(h->stream); hipStreamDestroy(h->stream); h->stream = nullptr; h->own_stream = false; }
    if (h->pinned) hipHostFree(h->pinned);
    if (h->pinned_rows) hipHostFree(h->pinned_rows);
    if (h->hstream && h->hstream != h->stream) { hipStreamSynchronize(h->hstream); hipStreamDestroy(h->hstream); }
    h->hstream = nullptr;
    for (int i = 0; i <= KNZ_STAGE_COUNT; i++) hipEventDestroy(h->ev[i]);
    if (h->pipe_ready) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); h->stream2 = nullptr; hipStreamSynchronize(h->stream3); hipStreamDestroy(h->stream3); h->stream3 = nullptr; }
    for (int i = 0; i < 3; i++) if (h->ev_pipe[i]) { hipEventDestroy(h->ev_pipe[i]); h->ev_pipe[i] = nullptr; }
    for (int i = 0; i < KNZ_MAX_PROBES; i++) if (h->probes[i].a) { hipEventDestroy(h->probes[i].a); hipEventDestroy(h->probes[i].b); }
    delete h;
    return KNZ_OK;
}

extern "C" const char* knz_last_error(void* handle) { return handle ? ((Handle*)handle)->err.c_str() : g_open_error.c_str(); }

static int multi_last_timing(Handle* h, float* stage_ms, int cap);
static int multi_last_counter(Handle* h, int id, uint64_t* value);

extern "C" int knz_last_timing(void* handle, float* stage_ms, int cap) {
    Handle* h = (Handle*)handle;
    if (!h || !stage_ms) return 0;
    if (h->multi) return multi_last_timing(h, stage_ms, cap);
    DeviceGuard dg(h);
    if (h->ev_valid) {
        hipEventSynchronize(h->ev[KNZ_STAGE_COUNT]);
        for (int i = 0; i < KNZ_STAGE_COUNT; i++) hipEventElapsedTime(&h->stage_ms[i], h->ev[i], h->ev[i + 1]);
    }
    int n = std::min(cap, (int)KNZ_STAGE_COUNT);
    for (int i = 0; i < n; i++) stage_ms[i] = h->stage_ms[i];
    return n;
}

extern "C" int knz_last_kernel_times(void* handle, char* names, int names_cap, float* ms, int cap) {
    Handle* h = lane0((Handle*)handle);                  // (several lanes: the launches of the first one)
    if (!h || !names || !ms || names_cap <= 0) return 0;
    DeviceGuard dg(h);
    int n = 0, pos = 0;
    names[0] = 0;
    for (int i = 0; i < h->nprobes && n < cap; i++) {
        const KernelProbe& p = h->probes[i];
        const int len = (int)strlen(p.name);
        if (pos + len + 2 > names_cap) break;
        if (hipEventSynchronize(p.b) != hipSuccess || hipEventElapsedTime(&ms[n], p.a, p.b) != hipSuccess) ms[n] = 0.f;
        memcpy(names + pos, p.name, len); pos += len; names[pos++] = '\n'; names[pos] = 0;
        n++;
    }
    return n;
}

extern "C" int knz_last_counter(void* handle, int id, uint64_t* value) {
    Handle* h = (Handle*)handle;
    if (!h || !value || id < KNZ_COUNTER_HUF_SERIAL_CHUNKS || id > KNZ_COUNTER_STAGE_BYTES0 + 7 || (id > KNZ_COUNTER_RANK_PIPE_BLOCKS && id < KNZ_COUNTER_STAGE_BYTES0)) return KNZ_ERR_INVALID_PARAM;
    if (h->multi) return multi_last_counter(h, id, value);
    DeviceGuard dg(h);
    *value = 0;
    if (id == KNZ_COUNTER_POST_TRANSFORM_BYTES) { *value = h->post_bytes; return KNZ_OK; }
    if (id >= KNZ_COUNTER_STAGE_BYTES0) { *value = h->stage_bytes[id - KNZ_COUNTER_STAGE_BYTES0]; return KNZ_OK; }
    if (id == KNZ_COUNTER_TEXT_CHAIN_BLOCKS) {
        uint32_t v = 0;
        if (h->text_cnt.p && hipMemcpy(&v, h->text_cnt.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return KNZ_ERR_UNKNOWN;
        *value = v;
        return KNZ_OK;
    }
    if (id == KNZ_COUNTER_LZ_FWD_ROUNDS) { *value = h->lzs_rounds; return KNZ_OK; }
    if (id == KNZ_COUNTER_RANK_PIPE_BLOCKS) {
        if (h->pipe_n == 0) return KNZ_OK;
        std::vector<uint8_t> f(h->pipe_n);
        if (hipMemcpy(f.data(), h->pipe_flag.p, f.size(), hipMemcpyDeviceToHost) != hipSuccess) return KNZ_ERR_UNKNOWN;
        uint64_t n = 0;
        for (uint8_t v : f) n += v;
        *value = n;
        return KNZ_OK;
    }
    if (id == KNZ_COUNTER_LZ_FWD_SERIAL_BLOCKS) {
        if (h->lzs_n == 0) return KNZ_OK;
        std::vector<uint8_t> f(h->lzs_n);
        if (hipMemcpy(f.data(), h->lzs_misc.p, f.size(), hipMemcpyDeviceToHost) != hipSuccess) return KNZ_ERR_UNKNOWN;
        uint64_t n = 0;
        for (uint8_t v : f) n += v == 2 ? 1 : 0;
        *value = n;
        return KNZ_OK;
    }
    if (id == KNZ_COUNTER_LZ_INV_SERIAL_BLOCKS) {
        if (h->lzi_serial_n == 0) return KNZ_OK;
        std::vector<uint8_t> f(h->lzi_serial_n);
        if (hipMemcpy(f.data(), h->lzi_serial.p, f.size(), hipMemcpyDeviceToHost) != hipSuccess) return KNZ_ERR_UNKNOWN;
        uint64_t n = 0;
        for (uint8_t v : f) n += v;
        *value = n;
        return KNZ_OK;
    }
    if (h->huf_fallback_n == 0) return KNZ_OK;
    std::vector<uint8_t> f(h->huf_fallback_n);
    if (hipMemcpy(f.data(), h->huf_fallback.p, f.size(), hipMemcpyDeviceToHost) != hipSuccess) return KNZ_ERR_UNKNOWN;
    uint64_t n = 0;
    for (uint8_t v : f) n += v;
    *value = n;
    return KNZ_OK;
}

#ifdef KNZ_PROFILE_PHASES
extern "C" int knz_debug_prof(unsigned long long* out, int reset) {
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_knz_prof), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_knz_prof), z, sizeof(z)); }
    return 0;
}
#endif

#include "knz_transforms.inc"

// ---- encode batch ----------------------------------------------------------------------------------------------------
// d_src: nblocks blocks, block b at b*block_size (last one shorter). Output either the framed .knz body/stream
// (framed=1) or per-block local streams at out_stride bytes (framed=0).
struct EncodeBatch {
    const uint8_t* d_src; uint64_t n;
    uint8_t* d_dst; uint64_t dst_cap;
    int framed; int with_header; int with_end; int64_t header_input_size;
    uint64_t out_stride;     // framed == 0
    int payload_only;        // 1: single EntropyEncoder object, no block header bits
    uint64_t total_bits;     // result
};

// block tables of an encode batch: absolute device addresses; blocks <= 15 bytes are copy blocks (CompressedStream.go:773-776)
struct EncTablesArgs {
    uint32_t nblocks; uint64_t src; uint64_t n; uint64_t bs; int payload_only; int none_only;
    uint64_t* blk_off; uint32_t* blk_len; uint32_t* blk_src_len; uint8_t* blk_skip; uint8_t* blk_copy; int32_t* blk_status;
    uint8_t* active; uint8_t* side;
};
__global__ void knz_enc_tables_kernel(EncTablesArgs a) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.nblocks) return;
    const uint64_t rest = a.n - (uint64_t)b * a.bs;
    const uint32_t len = (uint32_t)(rest < a.bs ? rest : a.bs);
    const bool copy = len <= 15 && !a.payload_only;
    a.blk_off[b] = a.src + (uint64_t)b * a.bs;
    a.blk_len[b] = len;
    a.blk_src_len[b] = a.payload_only ? (len > 16 ? len : 16u) : len;   // a bare EntropyEncoder has no copy-block rule
    a.blk_copy[b] = copy ? 1 : 0;
    a.blk_skip[b] = (copy || a.none_only) ? 0x7F : 0xFF;                // NullTransform always applies: slot 0 cleared
    a.blk_status[b] = 0;
    if (a.active) { a.active[b] = (copy || a.none_only) ? 0 : 1; a.side[b] = 0; }
}

// the rows of Handle::ResultRow for the blocks of a batch, the totals (bits written, overflow flag) in the row behind the last block
__global__ void knz_pack_results_kernel(uint32_t nblocks, const uint64_t* written, const uint64_t* cksum, const uint32_t* post_len, const int32_t* status,
                                        const uint32_t* hdr, const uint8_t* skip, const uint64_t* totals, Handle::ResultRow* rows) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < nblocks) {
        Handle::ResultRow r;
        r.written = written[b]; r.cksum = cksum[b]; r.post_len = post_len[b]; r.status = status[b]; r.mode = hdr[(size_t)b * 6 + 1]; r.skip = skip[b];
        rows[b] = r;
    } else if (b == nblocks) {
        Handle::ResultRow r;
        r.written = totals[0]; r.cksum = totals[1]; r.post_len = 0; r.status = 0; r.mode = 0; r.skip = 0;
        rows[b] = r;
    }
}

static int encode_batch(Handle* h, EncodeBatch& eb, hipStream_t st) {
    const knz_cfg& cfg = h->cfg;
    h->nprobes = 0;
    if (!transform_on_device(cfg.transform) || !entropy_on_device(cfg.entropy))
        return knz_set_error(h, KNZ_ERR_INVALID_CODEC, "transform/entropy combination has no device implementation in this build");
    if (eb.n == 0) {
        // Writer.Close on an empty stream: header + end marker only
        eb.total_bits = 0;
    }
    const uint64_t bs = cfg.block_size;
    const uint32_t nblocks = (uint32_t)((eb.n + bs - 1) / bs);
    const uint32_t chunkSize = (cfg.entropy == KNZ_E_ANS1 || cfg.entropy == KNZ_E_FPAQ) ? KNZ_ANS1_CHUNK : KNZ_HUF_CHUNK;
    const uint32_t maxPost = knz_max_encoded_len(cfg.transform, (uint32_t)std::min<uint64_t>(bs, eb.n ? eb.n : 1));
    const uint32_t cpb = std::max<uint32_t>(1, (maxPost + chunkSize - 1) / chunkSize);
    const size_t nslots = (size_t)std::max<uint32_t>(nblocks, 1) * cpb;
    const uint32_t slotStride = cfg.entropy == KNZ_E_ANS0 ? KNZ_ANS_SLOT : (cfg.entropy == KNZ_E_ANS1 ? KNZ_ANS1_SLOT : (cfg.entropy == KNZ_E_FPAQ ? KNZ_FPAQ_SLOT : KNZ_CHUNK_STRIDE));

    if (h->blk_off.reserve(sizeof(uint64_t) * (nblocks + 1)) || h->blk_len.reserve(4 * (nblocks + 1)) ||
        h->blk_src_len.reserve(4 * (nblocks + 1)) || h->blk_skip.reserve(nblocks + 16) || h->blk_cksum.reserve(8 * (nblocks + 1)) ||
        h->blk_status.reserve(4 * (nblocks + 1)) || h->unit_bits.reserve(4 * nslots * KNZ_UNITS_PER_CHUNK) || h->unit_src.reserve(4 * nslots * KNZ_UNITS_PER_CHUNK) ||
        h->scratch.reserve(nslots * (size_t)slotStride + 64) || h->ans_tab.reserve(cfg.entropy == KNZ_E_ANS0 ? nslots * 2048 + nslots * 4 : 16) || h->chunk_rel.reserve(8 * nslots) ||
        h->blk_written.reserve(8 * (nblocks + 1)) || h->blk_hdr.reserve(4 * 6 * (nblocks + 1)) ||
        h->blk_dst_bit.reserve(8 * (nblocks + 1)) || h->total_bits.reserve(64) || h->blk_copy.reserve(nblocks + 16))
        return knz_set_error(h, KNZ_ERR_CREATE_COMPRESSOR, "device workspace allocation failed");

    // block tables: absolute device addresses; blocks <= 15 bytes are copy blocks (CompressedStream.go:773-776)
    XfBatch xb;
    if (nblocks) {   // block tables, filled on the device: no staging copies, no host synchronisation in front of the first kernel
        const bool noneOnly = cfg.transform == 0;
        const uint64_t stride = ((uint64_t)maxPost + 64 + 15) & ~(uint64_t)15;
        if (!noneOnly && xf_alloc(h, xb, nblocks, stride)) return knz_set_error(h, KNZ_ERR_CREATE_COMPRESSOR, "device workspace allocation failed");
        EncTablesArgs ta;
        ta.nblocks = nblocks; ta.src = (uint64_t)eb.d_src; ta.n = eb.n; ta.bs = bs; ta.payload_only = eb.payload_only; ta.none_only = noneOnly ? 1 : 0;
        ta.blk_off = h->blk_off.as<uint64_t>(); ta.blk_len = h->blk_len.as<uint32_t>(); ta.blk_src_len = h->blk_src_len.as<uint32_t>();
        ta.blk_skip = h->blk_skip.as<uint8_t>(); ta.blk_copy = h->blk_copy.as<uint8_t>(); ta.blk_status = h->blk_status.as<int32_t>();
        ta.active = noneOnly ? nullptr : xb.active; ta.side = noneOnly ? nullptr : xb.side;
        hipLaunchKernelGGL(knz_enc_tables_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, st, ta);
    }
    const bool skipOpt = (cfg.flags & KNZ_FLAG_SKIP_BLOCKS) != 0 && !eb.payload_only && nblocks != 0;
    bool hufDirect = false;                                              // Huffman units encoded at their final bit positions (below)
    HufEncArgs hufArgs;
    if (skipOpt) {                                                       // -s: incompressible blocks become copy blocks (:778-800)
        SkipArgs ka;
        ka.nblocks = nblocks; ka.blk_off = h->blk_off.as<uint64_t>(); ka.blk_len = h->blk_len.as<uint32_t>();
        ka.blk_copy = h->blk_copy.as<uint8_t>(); ka.blk_skip = h->blk_skip.as<uint8_t>(); ka.active = cfg.transform != 0 ? xb.active : nullptr;
        hipLaunchKernelGGL(knz_skip_detect_kernel, dim3(nblocks), dim3(256), 0, st, ka);
    }
    hipEventRecord(h->ev[0], st);
    if (nblocks && cfg.checksum_bits != 0) {        // checksum of the untransformed block (encodingTask.encode :760-767)
        XxhArgs xa;
        xa.nblocks = nblocks; xa.ptr = h->blk_off.as<uint64_t>(); xa.len = h->blk_len.as<uint32_t>(); xa.cksum = h->blk_cksum.as<uint64_t>();
        xa.status = h->blk_status.as<int32_t>(); xa.mode = nullptr; xa.bits = cfg.checksum_bits; xa.verify = 0;
        hipLaunchKernelGGL(knz_xxhash_kernel, dim3(nblocks), dim3(64), 0, st, xa);
    }
    if (nblocks && cfg.transform != 0) {
        xb.cur_ptr = h->blk_off.as<uint64_t>(); xb.cur_len = h->blk_len.as<uint32_t>(); xb.skip = h->blk_skip.as<uint8_t>();
        xb.blk_status = h->blk_status.as<int32_t>();
        bool hasUtf = false;                                             // (or TEXT: the two stages that read and write ctx["dataType"])
        for (int sft = 42; sft >= 0; sft -= 6) hasUtf = hasUtf || ((cfg.transform >> sft) & 63) == KNZ_T_UTF || ((cfg.transform >> sft) & 63) == KNZ_T_TEXT;
        if (hasUtf) {                                                    // ctx["dataType"] from the magic number of the untransformed block (:811-819)
            if (h->blk_dt.reserve(nblocks + 16)) return knz_set_error(h, KNZ_ERR_CREATE_COMPRESSOR, "device workspace allocation failed");
            hipLaunchKernelGGL(knz_block_datatype_kernel, dim3((nblocks + 63) / 64), dim3(64), 0, st, nblocks, (const uint64_t*)h->blk_off.as<uint64_t>(),
                               (const uint32_t*)h->blk_len.as<uint32_t>(), h->blk_dt.as<uint8_t>());
            xb.blk_dt = h->blk_dt.as<uint8_t>();
        }
        int rc = forward_sequence(h, xb, cfg.transform, st);
        if (rc) return rc;
    }
    hipEventRecord(h->ev[1], st);
    if (nblocks) {
        if (cfg.entropy == KNZ_E_HUFFMAN || cfg.entropy == KNZ_E_NONE) {
            HufEncArgs a;
            a.data = nullptr; a.blk_off = h->blk_off.as<uint64_t>(); a.blk_len = h->blk_len.as<uint32_t>();
            a.chunks_per_block = cpb; a.scratch = h->scratch.as<uint8_t>(); a.unit_bits = h->unit_bits.as<uint32_t>(); a.unit_src = h->unit_src.as<uint32_t>();
            a.blk_status = h->blk_status.as<int32_t>();
            if (cfg.entropy == KNZ_E_HUFFMAN) {
                const uint32_t nc = nblocks * cpb, groups = (nc + 63) / 64;
                if (h->huf_stfreq.reserve((size_t)groups * 256 * 64 * 2) || h->huf_stsym.reserve((size_t)groups * 256 * 64) ||
                    h->huf_stlen.reserve((size_t)groups * 256 * 64) || h->huf_stcnt.reserve((size_t)groups * 64 * 2) || h->huf_stmax.reserve((size_t)groups * 64))
                    return knz_set_error(h, KNZ_ERR_CREATE_COMPRESSOR, "device workspace allocation failed");
                a.st_freq = h->huf_stfreq.as<uint16_t>(); a.st_sym = h->huf_stsym.as<uint8_t>(); a.st_len = h->huf_stlen.as<uint8_t>();
                a.st_count = h->huf_stcnt.as<uint16_t>(); a.st_maxlen = h->huf_stmax.as<uint8_t>(); a.nchunks = nc;
                // The units are encoded at their final bit positions (sizes pass -> layout scans -> encoder, further down) unless copy blocks of -s
                // have to overwrite chunks afterwards or the test switch asks for the scratch-slot form (units to slots, knz_gather_kernel).
                hufDirect = !skipOpt && knz_test_switch("KNZ_HUF_SCRATCH") == nullptr;
                a.st_fhist = nullptr; a.dst_words = nullptr; a.chunk_rel = nullptr; a.blk_dst_bit = nullptr; a.total_bits = nullptr;
                if (hufDirect) {
                    if (h->huf_fhist.reserve((size_t)nc * 4 * 256 * 2 + 64)) return knz_set_error(h, KNZ_ERR_CREATE_COMPRESSOR, "device workspace allocation failed");
                    a.st_fhist = h->huf_fhist.as<uint16_t>();
                }
                KNZ_LAUNCH_PROBED(knz_huf_hist_kernel, dim3(nc), dim3(256), 0, st, a);
                KNZ_LAUNCH_PROBED(knz_huf_lengths_kernel, dim3(groups), dim3(64), 0, st, a);
                if (hufDirect) { KNZ_LAUNCH_PROBED(knz_huf_encode_kernel<true>, dim3(nc), dim3(256), 0, st, a); hufArgs = a; }
                else KNZ_LAUNCH_PROBED(knz_huf_encode_kernel<false>, dim3(nc), dim3(256), 0, st, a);
            }
            else hipLaunchKernelGGL(knz_raw_units_kernel, dim3(nblocks * cpb), dim3(256), 0, st, a);
        } else if (cfg.entropy == KNZ_E_FPAQ) {
            FpaqArgs a;
            a.blk_off = h->blk_off.as<uint64_t>(); a.blk_len = h->blk_len.as<uint32_t>(); a.blk_src_len = h->blk_src_len.as<uint32_t>();
            a.chunks_per_block = cpb; a.scratch = h->scratch.as<uint8_t>(); a.unit_bits = h->unit_bits.as<uint32_t>();
            a.unit_src = h->unit_src.as<uint32_t>(); a.blk_status = h->blk_status.as<int32_t>();
            KNZ_LAUNCH_PROBED(knz_fpaq_encode_kernel, dim3(nblocks), dim3(64), 0, st, a);
        } else if (cfg.entropy == KNZ_E_ANS1) {
            // bounded groups of blocks (like the UTF stage and the suffix sort): a chunk slot takes 768 KiB of count / coder tables, 112 KiB of
            // context headers and the 64 MiB expanded-step stream; the workspace is sized to at most ~64 GiB of them (up to ~960 chunks side by side: the chains of a group run as one wave each, 25 ms whatever their number), not to the batch
            const size_t perSlot = (size_t)65536 * 12 + (size_t)256 * KNZ_ANS1_CTXHDR_BYTES + 1024 + KNZ_ANS1_ENT_STRIDE * 16;
            const uint32_t slotsPerGroup = (uint32_t)std::max<size_t>(cpb, std::min<size_t>((size_t)nblocks * cpb, ((size_t)64 << 30) / perSlot));
            uint32_t GB = std::max<uint32_t>(1, slotsPerGroup / cpb);                   // whole blocks per group
            if (const char* e = knz_test_switch("KNZ_ANS1_GROUP_BLOCKS")) GB = std::max(1, atoi(e));   // (tests: several groups on small inputs)
            const bool ans1EncPlain = knz_test_switch("KNZ_ANS1_ENC_PLAIN") != nullptr;   // (A/B and cross-check: the compiler's loop instead of the hand-written one)
            // a group the device has no room for (other handles, a smaller device) is halved until it fits: fewer chains side by side, same bytes
            uint32_t gs = GB * cpb;
            for (;;) {
                gs = GB * cpb;
                if (!(h->a1_freqs.reserve((size_t)gs * 65536 * 4) || h->a1_tab.reserve((size_t)gs * 65536 * 8) ||
                      h->a1_ctxhdr.reserve((size_t)gs * 256 * KNZ_ANS1_CTXHDR_BYTES + 64) || h->a1_ctxbits.reserve((size_t)gs * 256 * 4) ||
                      h->a1_ent.reserve((size_t)gs * KNZ_ANS1_ENT_STRIDE * 16)))
                    break;
                if (GB == 1) return knz_set_error(h, KNZ_ERR_CREATE_COMPRESSOR, "device workspace allocation failed");
                h->a1_freqs.release(); h->a1_tab.release(); h->a1_ctxhdr.release(); h->a1_ctxbits.release(); h->a1_ent.release();
                GB = (GB + 1) / 2;
            }
            for (uint32_t b0 = 0; b0 < nblocks; b0 += GB) {
                const uint32_t gb = std::min<uint32_t>(GB, nblocks - b0), ns = gb * cpb;
                const size_t s0 = (size_t)b0 * cpb;
                Ans1Args a;                                                                // the group's slice of every per-block / per-slot table
                a.blk_off = h->blk_off.as<uint64_t>() + b0; a.blk_len = h->blk_len.as<uint32_t>() + b0; a.chunks_per_block = cpb; a.nslots = ns;
                a.scratch = h->scratch.as<uint8_t>() + s0 * slotStride; a.unit_bits = h->unit_bits.as<uint32_t>() + s0 * 5; a.unit_src = h->unit_src.as<uint32_t>() + s0 * 5;
                a.freqs = h->a1_freqs.as<uint32_t>(); a.tab = h->a1_tab.as<uint2>(); a.ctx_hdr = h->a1_ctxhdr.as<uint8_t>();
                a.ctx_bits = h->a1_ctxbits.as<uint32_t>(); a.blk_status = h->blk_status.as<int32_t>() + b0;
                hipMemsetAsync(h->a1_freqs.p, 0, (size_t)ns * 65536 * 4, st);
                KNZ_LAUNCH_PROBED(knz_ans1_hist_kernel, dim3(ns * KNZ_ANS1_HIST_WGS * KNZ_ANS1_HIST_SLICES), dim3(256), 0, st, a);
                hipLaunchKernelGGL(knz_ans1_stats_kernel, dim3(ns * 256), dim3(64), 0, st, a);
                hipLaunchKernelGGL(knz_ans1_merge_kernel, dim3(ns), dim3(256), 0, st, a);
                KNZ_LAUNCH_PROBED(knz_ans1_expand_kernel, dim3(ns, 128), dim3(256), 0, st, a, h->a1_ent.as<uint4>());
#ifndef KNZ_HIP_EMU
                if (!ans1EncPlain) KNZ_LAUNCH_PROBED(knz_ans1_encode_asm_kernel, dim3(ns), dim3(64), 0, st, a, (const uint4*)h->a1_ent.as<uint4>());
                else
#endif
                KNZ_LAUNCH_PROBED(knz_ans1_encode_kernel, dim3(ns), dim3(64), 0, st, a, (const uint4*)h->a1_ent.as<uint4>());
            }
        } else if (cfg.entropy == KNZ_E_ANS0) {
            Ans0Args a;
            a.data = nullptr; a.blk_off = h->blk_off.as<uint64_t>(); a.blk_len = h->blk_len.as<uint32_t>();
            a.chunks_per_block = cpb; a.scratch = h->scratch.as<uint8_t>(); a.unit_bits = h->unit_bits.as<uint32_t>(); a.unit_src = h->unit_src.as<uint32_t>();
            a.tab = h->ans_tab.as<uint2>(); a.chunk_info = (uint32_t*)(h->ans_tab.as<uint8_t>() + nslots * 2048);
            a.blk_status = h->blk_status.as<int32_t>();
            const uint32_t ns = nblocks * cpb;
            hipLaunchKernelGGL(knz_ans0_stats_kernel, dim3(ns), dim3(256), 0, st, a);
            KNZ_LAUNCH_PROBED(knz_ans0_encode_kernel, dim3((ns + KNZ_ANS0_CHUNKS_PER_WG - 1) / KNZ_ANS0_CHUNKS_PER_WG), dim3(128), 0, st, a, ns);
        }
    }
    hipEventRecord(h->ev[2], st);
    if (skipOpt && cfg.entropy != KNZ_E_NONE) {
        CopyUnitsArgs ca;
        ca.chunks_per_block = cpb; ca.chunk_size = chunkSize; ca.blk_copy = h->blk_copy.as<uint8_t>(); ca.blk_off = h->blk_off.as<uint64_t>();
        ca.blk_len = h->blk_len.as<uint32_t>(); ca.scratch = h->scratch.as<uint8_t>(); ca.slot_stride = slotStride;
        ca.unit_bits = h->unit_bits.as<uint32_t>(); ca.unit_src = h->unit_src.as<uint32_t>(); ca.blk_status = h->blk_status.as<int32_t>();
        hipLaunchKernelGGL(knz_copy_units_kernel, dim3(nblocks * cpb), dim3(256), 0, st, ca);
    }
    LayoutArgs la;
    la.nblocks = nblocks; la.chunks_per_block = cpb; la.unit_bits = h->unit_bits.as<uint32_t>();
    la.blk_len = h->blk_len.as<uint32_t>(); la.blk_src_len = h->blk_src_len.as<uint32_t>(); la.blk_skip = h->blk_skip.as<uint8_t>(); la.blk_copy = h->blk_copy.as<uint8_t>();
    la.blk_cksum = h->blk_cksum.as<uint64_t>(); la.checksum_bits = cfg.checksum_bits; la.n_transforms = seq_len(cfg.transform);
    la.chunk_size = chunkSize; la.payload_only = eb.payload_only; la.chunk_rel = h->chunk_rel.as<uint64_t>(); la.blk_written = h->blk_written.as<uint64_t>();
    la.blk_hdr = h->blk_hdr.as<uint32_t>();
    if (nblocks) hipLaunchKernelGGL(knz_layout_blocks_kernel, dim3(nblocks), dim3(256), 0, st, la);

    StreamArgs sa;
    sa.nblocks = nblocks; sa.chunks_per_block = cpb; sa.chunk_size = chunkSize; sa.blk_len = h->blk_len.as<uint32_t>();
    sa.chunk_rel = h->chunk_rel.as<uint64_t>(); sa.blk_written = h->blk_written.as<uint64_t>(); sa.blk_hdr = h->blk_hdr.as<uint32_t>();
    sa.dst_words = (uint32_t*)eb.d_dst;
    const uint64_t usable = eb.dst_cap >= 8 ? ((eb.dst_cap & ~(uint64_t)3) - 4) : 0;   // whole BE words are stored
    sa.dst_cap_bits = usable * 8;
    sa.first_bit = 0; sa.framed = eb.framed; sa.block_stride_bits = eb.out_stride * 8; sa.end_marker = eb.with_end;
    sa.header_bits = 0;
    for (int i = 0; i < 8; i++) sa.header_words[i] = 0;
    if (eb.framed && eb.with_header) sa.header_bits = knz_build_stream_header(cfg, eb.header_input_size, sa.header_words);
    sa.blk_dst_bit = h->blk_dst_bit.as<uint64_t>(); sa.total_bits = h->total_bits.as<uint64_t>();
    sa.blk_status = h->blk_status.as<int32_t>();
    hipLaunchKernelGGL(knz_layout_stream_kernel, dim3(1), dim3(256), 0, st, sa);
    hipEventRecord(h->ev[3], st);
    if (hufDirect && nblocks) {                                          // Huffman: the encoder itself places the units (no scratch round trip, no gather)
        hufArgs.dst_words = (uint32_t*)eb.d_dst; hufArgs.chunk_rel = h->chunk_rel.as<uint64_t>(); hufArgs.blk_dst_bit = h->blk_dst_bit.as<uint64_t>();
        hufArgs.total_bits = h->total_bits.as<uint64_t>();
        KNZ_LAUNCH_PROBED(knz_huf_encode_kernel<false>, dim3(nblocks * cpb), dim3(256), 0, st, hufArgs);
    }

    GatherArgs ga;
    ga.chunks_per_block = cpb; ga.chunk_size = chunkSize; ga.blk_len = h->blk_len.as<uint32_t>(); ga.unit_bits = h->unit_bits.as<uint32_t>();
    ga.scratch = h->scratch.as<uint8_t>(); ga.chunk_stride = slotStride;
    ga.unit_src = h->unit_src.as<uint32_t>();
    ga.chunk_rel = h->chunk_rel.as<uint64_t>(); ga.blk_dst_bit = h->blk_dst_bit.as<uint64_t>(); ga.dst_words = (uint32_t*)eb.d_dst;
    ga.total_bits = h->total_bits.as<uint64_t>();
    if (nblocks && !hufDirect) KNZ_LAUNCH_PROBED(knz_gather_kernel, dim3(nblocks * cpb, (cfg.entropy == KNZ_E_ANS1 || cfg.entropy == KNZ_E_FPAQ) ? 64 : 1), dim3(256), 0, st, ga);
    hipEventRecord(h->ev[4], st);
    h->ev_valid = true;

    // results come back packed: one row per block (bit count, checksum, post-transform length, status, mode, skip flags) and the batch totals, gathered
    // by one small kernel and brought over by ONE asynchronous copy into pinned memory, one synchronisation
    if (h->res_rows.reserve(sizeof(Handle::ResultRow) * ((size_t)nblocks + 1)) || h->reserve_pinned_rows((size_t)nblocks))
        return knz_set_error(h, KNZ_ERR_CREATE_COMPRESSOR, "pinned host allocation failed");
    hipLaunchKernelGGL(knz_pack_results_kernel, dim3((nblocks + 1 + 255) / 256), dim3(256), 0, st, nblocks, (const uint64_t*)h->blk_written.as<uint64_t>(),
                       (const uint64_t*)h->blk_cksum.as<uint64_t>(), (const uint32_t*)h->blk_len.as<uint32_t>(), (const int32_t*)h->blk_status.as<int32_t>(),
                       (const uint32_t*)h->blk_hdr.as<uint32_t>(), (const uint8_t*)h->blk_skip.as<uint8_t>(), (const uint64_t*)h->total_bits.as<uint64_t>(),
                       h->res_rows.as<Handle::ResultRow>());
    Handle::ResultRow* rows = h->pinned_rows;
    HIP_OK(hipMemcpyAsync(rows, h->res_rows.p, sizeof(Handle::ResultRow) * ((size_t)nblocks + 1), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    HIP_OK(hipGetLastError());
    if (rows[nblocks].cksum != 0) return knz_set_error(h, KNZ_ERR_WRITE_FILE, "destination buffer too small");     // (the totals row: bits, overflow flag)
    for (uint32_t b = 0; b < nblocks; b++)
        if (rows[b].status != 0) return knz_set_error(h, rows[b].status, "block failed (the reference panics on this input: ERR_PROCESS_BLOCK)");
    eb.total_bits = rows[nblocks].written;
    h->post_bytes = 0;
    for (uint32_t b = 0; b < nblocks; b++) h->post_bytes += rows[b].post_len;
    for (int i = 0; i < 8; i++) h->stage_bytes[i] = (nblocks && cfg.transform != 0) ? ((const uint64_t*)((const uint8_t*)h->pinned + 3072))[i] : 0;
    return KNZ_OK;
}

extern "C" int knz_dev_compress(void* handle, const void* d_src, uint64_t n, int64_t header_input_size, void* d_dst,
                                uint64_t dst_cap, uint64_t* out_bytes, void* hip_stream) {
    Handle* h = lane_of_pointer((Handle*)handle, d_dst);
    if (!h || !d_dst || !out_bytes || (!d_src && n)) return KNZ_ERR_MISSING_PARAM;
    DeviceGuard dg(h);
    if (((uintptr_t)d_dst & 3) || ((uintptr_t)d_src & 15)) return knz_set_error(h, KNZ_ERR_INVALID_PARAM, "d_src must be 16-byte and d_dst 4-byte aligned");
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;
    EncodeBatch eb{(const uint8_t*)d_src, n, (uint8_t*)d_dst, dst_cap, 1, 1, 1, header_input_size, 0, 0, 0};
    int rc = encode_batch(h, eb, st);
    if (rc) return rc;
    *out_bytes = (eb.total_bits + 7) >> 3;
    return KNZ_OK;
}

extern "C" int knz_dev_compress_blocks(void* handle, const void* d_src, uint64_t n, void* d_dst, uint64_t dst_cap,
                                       uint64_t* out_bits, void* hip_stream) {
    Handle* h = lane_of_pointer((Handle*)handle, d_dst);
    if (!h || !d_dst || !out_bits || (!d_src && n)) return KNZ_ERR_MISSING_PARAM;
    DeviceGuard dg(h);
    if (((uintptr_t)d_dst & 3) || ((uintptr_t)d_src & 15)) return knz_set_error(h, KNZ_ERR_INVALID_PARAM, "d_src must be 16-byte and d_dst 4-byte aligned");
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;
    EncodeBatch eb{(const uint8_t*)d_src, n, (uint8_t*)d_dst, dst_cap, 1, 0, 0, 0, 0, 0, 0};
    int rc = encode_batch(h, eb, st);
    if (rc) return rc;
    *out_bits = eb.total_bits;
    return KNZ_OK;
}

#include "knz_host_api.inc"
#include "knz_multi.inc"
