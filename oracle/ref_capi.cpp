// oracle/_ref — TEST INFRASTRUCTURE ONLY. Not part of the product path.
//
// C entry points over the kanzi-go sources themselves, translated mechanically to C++ by tools/go2cpp (`make -C oracle _ref`
// writes oracle/_ref/kanzi_ref.gen.hpp from /root/reference/v2/...; the generated file is never edited and never committed).
// This file is the only hand-written code in front of the generated one: it builds the objects the way the reference's own
// callers do (entropy.NewEntropyEncoder / transform.New and io/CompressedStream.go:804-914,1943-1990), hands them byte buffers
// and copies the results out. An in-memory io.WriteCloser / io.ReadCloser stands in for the file the bit stream writes to.
//
// What it is for: pinning the hand-written oracle (oracle/*.hpp) and the device against the reference's OWN code:
// tests/test_ref_build.py (CPU) and tests/test_parity_gpu.py::test_device_vs_ref_* (GPU).
#ifdef KREF_WITH_CGO
// oracle/_ref/libknz_ref_gpu.so: the same translation PLUS the cgo shim of go/ (gpu_batch.go, gpu_stream.go, gpu_transform.go, gpu_entropy.go) and the
// two-line patch of INTEGRATION.md in CompressedStream.go, linked against libknz_gpu.so: the reference's Writer / Reader drive the device.
#include "kanzi_ref_gpu.gen.hpp"
#else
#include "kanzi_ref.gen.hpp"
#endif

#include <string>

namespace {

thread_local std::string g_err;

// in-memory stream under DefaultOutputBitStream / DefaultInputBitStream (the reference tests use internal.BufferStream = bytes.Buffer)
struct MemStream : go_io::WriteCloser, go_io::ReadCloser {
    std::string data;
    size_t rd = 0;
    std::tuple<go::Int, go::error> Write(go::Slice<go::Byte> b) override {
        data.append((const char*)b.p, (size_t)b.n);
        return {go::Int::from_raw(b.n), nullptr};
    }
    std::tuple<go::Int, go::error> Read(go::Slice<go::Byte> b) override {
        size_t n = std::min<size_t>((size_t)b.n, data.size() - rd);
        if (n == 0 && b.n > 0) return {go::Int(), go_io::EOF_};
        memcpy((void*)b.p, data.data() + rd, n);
        rd += n;
        return {go::Int::from_raw((int64_t)n), nullptr};
    }
    go::error Close() override { return nullptr; }
};

go::Slice<go::Byte> view(const uint8_t* p, uint64_t n) { return go::Slice<go::Byte>((go::Byte*)p, (int64_t)n, (int64_t)n); }
go::Slice<go::Byte> copy_in(const uint8_t* p, uint64_t n, uint64_t cap = 0) {
    go::Slice<go::Byte> s = go::Slice<go::Byte>::make((int64_t)n, (int64_t)std::max(n, cap));
    if (n) memcpy((void*)s.p, p, n);
    return s;
}

thread_local uint32_t tlsBlockSize = 0, tlsEntropy = 0xFFFFFFFFu;   // ctx["blockSize"] / ctx["entropy"]: 0 / 0xFFFFFFFF = key absent
thread_local int tlsDataType = -1;                                 // ctx["dataType"]: -1 = key absent
thread_local unsigned tlsBsVersion = 6;

using Ctx = go::Map<go::String, go::any>;
const char* entropy_name(uint32_t t) {
    switch (t) { case 0: return "NONE"; case 1: return "HUFFMAN"; case 2: return "FPAQ"; case 3: return "PAQ"; case 4: return "RANGE"; case 5: return "ANS0";
                 case 6: return "CM"; case 7: return "TPAQ"; case 8: return "ANS1"; case 9: return "TPAQX"; default: return "?"; }
}
// the keys io/CompressedStream.go puts into the ctx of a block task (:218-224, :382) as far as the codecs read them
Ctx make_ctx() {
    Ctx c = go::make_map<go::String, go::any>();
    c[go::String("bsVersion")] = go::any(go::Uint(go::U(tlsBsVersion)));
    c[go::String("jobs")] = go::any(go::Uint(go::U(1)));
    if (tlsBlockSize) c[go::String("blockSize")] = go::any(go::Uint(go::U(tlsBlockSize)));
    if (tlsEntropy != 0xFFFFFFFFu) c[go::String("entropy")] = go::any(go::String(entropy_name(tlsEntropy)));
    if (tlsDataType >= 0) c[go::String("dataType")] = go::any(kz_internal::DataType(go::U(tlsDataType)));
    return c;
}
void read_back_data_type(Ctx& c) {
    auto [v, ok] = go::map_get2(c, go::String("dataType"));
    if (ok) { auto [dt, ok2] = go::assert2<kz_internal::DataType>(v); if (ok2) tlsDataType = (int)dt.v; }
}

#define KREF_TRY go::ArenaScope arena_; try {
#define KREF_CATCH                                                                         \
    }                                                                                      \
    catch (const go::PanicException& e) { g_err = std::string("panic: ") + e.what(); return 3; /* kanzi.ERR_PROCESS_BLOCK: a recovered panic */ } \
    catch (const std::exception& e) { g_err = e.what(); return 127; }

int fail(go::error e, int rc) { g_err = e ? e->Error().s : std::string("error"); return rc; }

}  // namespace

extern "C" {

const char* kref_last_error() { return g_err.c_str(); }
int kref_set_bs_version(unsigned v) { tlsBsVersion = v; return 0; }

// == entropy.NewEntropyEncoder(obs, ctx, type); ee.Write(block); ee.Dispose(); obs.Close()  (io/CompressedStream.go:898-914)
int kref_entropy_encode(uint32_t type, const uint8_t* src, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_bits) {
    KREF_TRY
    MemStream ms;
    auto [obs, err0] = kz_bitstream::NewDefaultOutputBitStream(&ms, go::Uint(go::U(16384)));
    if (err0 != nullptr) return fail(err0, 1);
    auto [ee, err] = kz_entropy::NewEntropyEncoder(obs, make_ctx(), go::Uint32(go::U(type)));
    if (err != nullptr) return fail(err, 2);
    auto [wr, werr] = ee->Write(copy_in(src, n));
    if (werr != nullptr) return fail(werr, 3);
    ee->Dispose();
    obs->Close();
    uint64_t bits = obs->Written().v;
    if (ms.data.size() > cap) { g_err = "output buffer too small"; return 4; }
    memcpy(out, ms.data.data(), ms.data.size());
    *out_bits = bits;
    return 0;
    KREF_CATCH
}

// == entropy.NewEntropyDecoder(ibs, ctx, type); ed.Read(buffer[0:n]); ed.Dispose()  (io/CompressedStream.go:1943-1965)
int kref_entropy_decode(uint32_t type, const uint8_t* bits, uint64_t nbytes, uint8_t* dst, uint64_t n, uint64_t* used_bits) {
    KREF_TRY
    MemStream ms;
    ms.data.assign((const char*)bits, (size_t)nbytes);
    auto [ibs, err0] = kz_bitstream::NewDefaultInputBitStream(&ms, go::Uint(go::U(16384)));
    if (err0 != nullptr) return fail(err0, 1);
    auto [ed, err] = kz_entropy::NewEntropyDecoder(ibs, make_ctx(), go::Uint32(go::U(type)));
    if (err != nullptr) return fail(err, 2);
    go::Slice<go::Byte> buf = go::Slice<go::Byte>::make((int64_t)n, (int64_t)n);
    auto [rd, rerr] = ed->Read(buf);
    if (rerr != nullptr) return fail(rerr, 3);
    ed->Dispose();
    if (n) memcpy(dst, buf.p, n);
    if (used_bits) *used_bits = ibs->Read().v;
    return 0;
    KREF_CATCH
}

// ctx of the calling thread for the entry points below (block_size 0 / entropy 0xFFFFFFFF / data_type -1 = key absent)
int kref_set_ctx(uint32_t block_size, uint32_t entropy_type) { tlsBlockSize = block_size; tlsEntropy = entropy_type; return 0; }
int kref_set_data_type(int dt) { tlsDataType = dt; return 0; }
int kref_get_data_type() { return tlsDataType; }

// one transform object: transform.newToken(ctx, t).Forward(src, dst)   (t = the 6-bit id, transform/Factory.go:31-53). rc -1 = Forward returned
// an error (= "skip me" for the sequence)
int kref_transform_forward(uint64_t t, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KREF_TRY
    Ctx ctx = make_ctx();
    auto [tr, err] = kz_transform::newToken(&ctx, go::Uint64(go::U(t)));
    if (err != nullptr) return fail(err, 2);
    go::Slice<go::Byte> in = copy_in(src, n);
    go::Slice<go::Byte> outb = go::Slice<go::Byte>::make((int64_t)cap, (int64_t)cap);
    auto [rd, wr, ferr] = tr->Forward(in, outb);
    read_back_data_type(ctx);
    if (ferr != nullptr) { fail(ferr, -1); return -1; }
    if (wr.v) memcpy(dst, outb.p, wr.v);
    *out_n = wr.v;
    return 0;
    KREF_CATCH
}

int kref_transform_inverse(uint64_t t, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KREF_TRY
    Ctx ctx = make_ctx();
    auto [tr, err] = kz_transform::newToken(&ctx, go::Uint64(go::U(t)));
    if (err != nullptr) return fail(err, 2);
    go::Slice<go::Byte> in = copy_in(src, n);
    go::Slice<go::Byte> outb = go::Slice<go::Byte>::make((int64_t)cap, (int64_t)cap);
    auto [rd, wr, ierr] = tr->Inverse(in, outb);
    if (ierr != nullptr) return fail(ierr, 3);
    if (wr.v) memcpy(dst, outb.p, wr.v);
    *out_n = wr.v;
    return 0;
    KREF_CATCH
}

uint64_t kref_max_encoded_len(uint64_t type, uint64_t n) {
    go::ArenaScope arena_;
    try {
        Ctx ctx = make_ctx();
        auto [seq, err] = kz_transform::New(&ctx, go::Uint64(go::U(type)));
        if (err != nullptr) return 0;
        return (uint64_t)seq->MaxEncodedLen(go::Int(go::U(n))).v;
    } catch (...) { return 0; }
}

// transform.New(ctx, type).Forward(src, dst) + SkipFlags()   (io/CompressedStream.go:804-834)
int kref_sequence_forward(uint64_t type, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* out_n, uint8_t* skip_flags) {
    KREF_TRY
    Ctx ctx = make_ctx();
    auto [seq, err] = kz_transform::New(&ctx, go::Uint64(go::U(type)));
    if (err != nullptr) return fail(err, 2);
    uint64_t req = (uint64_t)seq->MaxEncodedLen(go::Int(go::U(n))).v;
    go::Slice<go::Byte> in = copy_in(src, n, req);
    go::Slice<go::Byte> outb = go::Slice<go::Byte>::make((int64_t)std::max(req, cap), (int64_t)std::max(req, cap));
    auto [rd, wr, ferr] = seq->Forward(in, outb);
    read_back_data_type(ctx);
    if (ferr != nullptr) return fail(ferr, 3);
    if (wr.v > cap) { g_err = "output buffer too small"; return 4; }
    if (wr.v) memcpy(dst, outb.p, wr.v);
    *out_n = wr.v;
    *skip_flags = seq->SkipFlags().v;
    return 0;
    KREF_CATCH
}

// transform.New(ctx, type); SetSkipFlags; Inverse   (io/CompressedStream.go:1974-1990)
int kref_sequence_inverse(uint64_t type, uint8_t skip_flags, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KREF_TRY
    Ctx ctx = make_ctx();
    auto [seq, err] = kz_transform::New(&ctx, go::Uint64(go::U(type)));
    if (err != nullptr) return fail(err, 2);
    seq->SetSkipFlags(go::Byte(go::U(skip_flags)));
    go::Slice<go::Byte> in = copy_in(src, n, cap);
    go::Slice<go::Byte> outb = go::Slice<go::Byte>::make((int64_t)cap, (int64_t)cap);
    auto [rd, wr, ierr] = seq->Inverse(in, outb);
    if (ierr != nullptr) return fail(ierr, 3);
    if (wr.v) memcpy(dst, outb.p, wr.v);
    *out_n = wr.v;
    return 0;
    KREF_CATCH
}

// BWT object on its own: bytes + the 8 primary indexes (transform/BWT.go:132-209, :178)
int kref_bwt_forward(const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t* primary8) {
    KREF_TRY
    auto [bwt, err] = kz_transform::NewBWT();
    if (err != nullptr) return fail(err, 2);
    go::Slice<go::Byte> outb = go::Slice<go::Byte>::make((int64_t)n, (int64_t)n);
    auto [rd, wr, ferr] = bwt->Forward(copy_in(src, n), outb);
    if (ferr != nullptr) return fail(ferr, 3);
    if (n) memcpy(dst, outb.p, n);
    for (int i = 0; i < 8; i++) primary8[i] = bwt->PrimaryIndex(go::Int(go::U(i))).v;
    return 0;
    KREF_CATCH
}

int kref_bwt_inverse(const uint8_t* src, uint64_t n, uint8_t* dst, const uint64_t* primary8) {
    KREF_TRY
    auto [bwt, err] = kz_transform::NewBWT();
    if (err != nullptr) return fail(err, 2);
    for (int i = 0; i < 8; i++) bwt->SetPrimaryIndex(go::Int(go::U(i)), go::Uint(go::U(primary8[i])));
    go::Slice<go::Byte> outb = go::Slice<go::Byte>::make((int64_t)n, (int64_t)n);
    auto [rd, wr, ierr] = bwt->Inverse(copy_in(src, n), outb);
    if (ierr != nullptr) return fail(ierr, 3);
    if (n) memcpy(dst, outb.p, n);
    return 0;
    KREF_CATCH
}

// A kanzi.Listener (Definitions.go, Event.go) that writes down what it is told: "type id size hash hashType" or "type id msg" per event. When
// kref_record_events(verbosity) has been called, the four stream calls below attach it (Writer.AddListener / Reader.AddListener) and put
// ctx["verbosity"] in place; kref_event_log returns the lines of the last call. Times are left out: they are the only field that may differ.
struct EventRecorder : kz_kanzi::Listener {
    std::string log;
    void ProcessEvent(kz_kanzi::Event* e) override {
        char line[160];
        if (e->msg.s.empty())
            snprintf(line, sizeof line, "%lld %lld %lld %llx %lld\n", (long long)e->eventType.v, (long long)e->id.v, (long long)e->size.v, (unsigned long long)e->hash.v,
                     (long long)e->hashType.v);
        else
            snprintf(line, sizeof line, "%lld %lld %s\n", (long long)e->eventType.v, (long long)e->id.v, e->msg.s.c_str());
        log += line;
    }
};
static int g_record_verbosity = -1;          // < 0: no listener
static thread_local std::string g_event_log;   // (per calling thread: bench.py's CPU baseline runs many of these calls side by side)
extern "C" void kref_record_events(int verbosity) { g_record_verbosity = verbosity; g_event_log.clear(); }
extern "C" uint64_t kref_event_log(char* dst, uint64_t cap) {
    if (dst && cap) { size_t k = std::min<size_t>(g_event_log.size(), (size_t)cap - 1); memcpy(dst, g_event_log.data(), k); dst[k] = 0; }
    return g_event_log.size();
}

// == what app/BlockCompressor.go does with a file: io.NewWriterWithCtx(os, ctx); w.Write(data); w.Close()  (io/CompressedStream.go:232-620).
// The goroutines of Writer.processBlock run one after the other (`go task.encode(..)` is emitted as the call; the tasks' lock-free hand-over of
// the shared bit stream, :935-949, is satisfied in block order). file_size < 0: ctx["fileSize"] absent. Returns the .knz stream.
int kref_compress(const uint8_t* src, uint64_t n, const char* transform, const char* entropy, uint32_t block_size, uint32_t checksum_bits, uint32_t jobs,
                  int64_t file_size, int skip_blocks, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KREF_TRY
    MemStream ms;
    Ctx ctx = go::make_map<go::String, go::any>();
    ctx[go::String("entropy")] = go::any(go::String(entropy));
    ctx[go::String("transform")] = go::any(go::String(transform));
    ctx[go::String("blockSize")] = go::any(go::Uint(go::U(block_size)));
    ctx[go::String("jobs")] = go::any(go::Uint(go::U(jobs ? jobs : 1)));
    ctx[go::String("checksum")] = go::any(go::Uint(go::U(checksum_bits)));
    if (file_size >= 0) ctx[go::String("fileSize")] = go::any(go::Int64(go::U(file_size)));
    ctx[go::String("headerless")] = go::any(false);
    if (skip_blocks) ctx[go::String("skipBlocks")] = go::any(true);
    if (g_record_verbosity >= 0) ctx[go::String("verbosity")] = go::any(go::Uint(go::U(g_record_verbosity)));
    auto [w, err] = kz_io::NewWriterWithCtx(&ms, ctx);
    if (err != nullptr) return fail(err, 1);
    EventRecorder rec;
    if (g_record_verbosity >= 0) w->AddListener(&rec);
    // (the application hands the writer its read buffer piece by piece; one call with everything writes the same stream)
    auto [wr, werr] = w->Write(copy_in(src, n));
    if (werr != nullptr) return fail(werr, 2);
    go::error cerr = w->Close();
    g_event_log = rec.log;
    if (cerr != nullptr) return fail(cerr, 3);
    if (ms.data.size() > cap) { g_err = "output buffer too small"; return 4; }
    memcpy(dst, ms.data.data(), ms.data.size());
    *out_n = ms.data.size();
    return 0;
    KREF_CATCH
}

// io.NewReader(is, jobs), or io.NewReaderWithCtx with ctx["verbosity"] when events are being recorded
// kref_decode_range(from, to): ctx["from"] / ctx["to"] of the readers opened from now on (block ids, first block = 1; the CLI's --from / --to); < 0 = key absent
static int g_from = -1, g_to = -1;
extern "C" void kref_decode_range(int from, int to) { g_from = from; g_to = to; }
static std::tuple<kz_io::Reader*, go::error> open_reader(MemStream* ms, uint32_t jobs) {
    if (g_record_verbosity < 0 && g_from < 0 && g_to < 0) return kz_io::NewReader(ms, go::Uint(go::U(jobs ? jobs : 1)));
    Ctx ctx = go::make_map<go::String, go::any>();
    ctx[go::String("jobs")] = go::any(go::Uint(go::U(jobs ? jobs : 1)));
    if (g_from >= 0) ctx[go::String("from")] = go::any(go::Int(go::U(g_from)));
    if (g_to >= 0) ctx[go::String("to")] = go::any(go::Int(go::U(g_to)));
    if (g_record_verbosity >= 0) ctx[go::String("verbosity")] = go::any(go::Uint(go::U(g_record_verbosity)));
    return kz_io::NewReaderWithCtx(ms, ctx);
}

// == io.NewReader(is, jobs); r.Read(...) until EOF; r.Close()  (io/CompressedStream.go:1047-1760)
int kref_decompress(const uint8_t* src, uint64_t n, uint32_t jobs, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KREF_TRY
    MemStream ms;
    ms.data.assign((const char*)src, (size_t)n);
    auto [r, err] = open_reader(&ms, jobs);
    if (err != nullptr) return fail(err, 1);
    EventRecorder rec;
    if (g_record_verbosity >= 0) r->AddListener(&rec);
    struct KeepLog { EventRecorder& r; ~KeepLog() { g_event_log = r.log; } } keep{rec};
    go::Slice<go::Byte> buf = go::Slice<go::Byte>::make(1 << 20, 1 << 20);
    uint64_t total = 0;
    while (true) {
        auto [k, rerr] = r->Read(buf);
        if (k.v > 0) {
            if (total + (uint64_t)k.v > cap) { g_err = "output buffer too small"; return 4; }
            memcpy(dst + total, buf.p, (size_t)k.v);
            total += (uint64_t)k.v;
        }
        if (rerr != nullptr) {
            if (rerr == go_io::EOF_) break;
            return fail(rerr, 2);
        }
        if (k.v == 0) break;
    }
    r->Close();
    *out_n = total;
    return 0;
    KREF_CATCH
}

#ifdef KREF_WITH_CGO
// kref_gpu_depth(n): from now on the two calls below ask for a batch depth of their own (Writer / Reader.EnableGPUDepth, go/gpu_stream.go); 0 = `jobs` blocks per batch
static int g_gpu_depth = 0;
extern "C" void kref_gpu_depth(int depth) { g_gpu_depth = depth; }
// kref_gpu_lanes(n): from now on the two calls below open the scheduler over n lanes (Writer / Reader.EnableGPUDevices with n times ordinal 0: logical
// devices on the one GPU of the box); 0 = the one-device handle of EnableGPU / EnableGPUDepth
static int g_gpu_lanes = 0;
extern "C" void kref_gpu_lanes(int lanes) { g_gpu_lanes = lanes; }
extern "C++" template <class W> go::error enable_gpu(W* w) {
    if (g_gpu_lanes > 0) {
        go::Slice<go::Int> devs = go::Slice<go::Int>::make(g_gpu_lanes, g_gpu_lanes);   // (all zero: ordinal 0)
        return w->EnableGPUDevices(devs, go::Int(go::U(g_gpu_depth)));
    }
    return g_gpu_depth > 0 ? w->EnableGPUDepth(go::Int(go::U(g_gpu_depth))) : w->EnableGPU();
}

// The Go host with its block batches re-pointed at the GPU batch scheduler: io.NewWriterWithCtx + Writer.EnableGPU (go/gpu_stream.go), `jobs` blocks per
// device batch. Same arguments and result as kref_compress.
int kref_gpu_compress(const uint8_t* src, uint64_t n, const char* transform, const char* entropy, uint32_t block_size, uint32_t checksum_bits, uint32_t jobs,
                      int64_t file_size, int skip_blocks, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KREF_TRY
    MemStream ms;
    Ctx ctx = go::make_map<go::String, go::any>();
    ctx[go::String("entropy")] = go::any(go::String(entropy));
    ctx[go::String("transform")] = go::any(go::String(transform));
    ctx[go::String("blockSize")] = go::any(go::Uint(go::U(block_size)));
    ctx[go::String("jobs")] = go::any(go::Uint(go::U(jobs ? jobs : 1)));
    ctx[go::String("checksum")] = go::any(go::Uint(go::U(checksum_bits)));
    if (file_size >= 0) ctx[go::String("fileSize")] = go::any(go::Int64(go::U(file_size)));
    ctx[go::String("headerless")] = go::any(false);
    if (skip_blocks) ctx[go::String("skipBlocks")] = go::any(true);
    if (g_record_verbosity >= 0) ctx[go::String("verbosity")] = go::any(go::Uint(go::U(g_record_verbosity)));
    auto [w, err] = kz_io::NewWriterWithCtx(&ms, ctx);
    if (err != nullptr) return fail(err, 1);
    EventRecorder rec;
    if (g_record_verbosity >= 0) w->AddListener(&rec);
    go::error gerr = enable_gpu(w);
    if (gerr != nullptr) return fail(gerr, 5);
    auto [wr, werr] = w->Write(copy_in(src, n));
    go::error cerr = werr != nullptr ? werr : w->Close();
    w->DisableGPU();
    g_event_log = rec.log;
    if (cerr != nullptr) return fail(cerr, 3);
    if (ms.data.size() > cap) { g_err = "output buffer too small"; return 4; }
    memcpy(dst, ms.data.data(), ms.data.size());
    *out_n = ms.data.size();
    return 0;
    KREF_CATCH
}

// io.NewReader + Reader.EnableGPU: the payloads are read from the stream by the Go code, decoded by the device batch
int kref_gpu_decompress(const uint8_t* src, uint64_t n, uint32_t jobs, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KREF_TRY
    MemStream ms;
    ms.data.assign((const char*)src, (size_t)n);
    auto [r, err] = open_reader(&ms, jobs);
    if (err != nullptr) return fail(err, 1);
    EventRecorder rec;
    if (g_record_verbosity >= 0) r->AddListener(&rec);
    struct KeepLog { EventRecorder& r; ~KeepLog() { g_event_log = r.log; } } keep{rec};
    go::error gerr = enable_gpu(r);
    if (gerr != nullptr) return fail(gerr, 5);
    go::Slice<go::Byte> buf = go::Slice<go::Byte>::make(1 << 20, 1 << 20);
    uint64_t total = 0;
    int rc = 0;
    while (true) {
        auto [k, rerr] = r->Read(buf);
        if (k.v > 0) {
            if (total + (uint64_t)k.v > cap) { g_err = "output buffer too small"; rc = 4; break; }
            memcpy(dst + total, buf.p, (size_t)k.v);
            total += (uint64_t)k.v;
        }
        if (rerr != nullptr) {
            if (rerr == go_io::EOF_) break;
            rc = fail(rerr, 2);
            break;
        }
        if (k.v == 0) break;
    }
    r->Close();
    r->DisableGPU();
    if (rc) return rc;
    *out_n = total;
    return 0;
    KREF_CATCH
}

// kanzi.ByteTransform / kanzi.EntropyEncoder / kanzi.EntropyDecoder objects of go/gpu_transform.go and go/gpu_entropy.go over a handle of knz_open
static void* open_handle(uint32_t block_size) {
    ::knz_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.block_size = block_size; cfg.bs_version = 6; cfg.device = -1;
    void* h = nullptr;
    return ::knz_open(&cfg, &h) == 0 ? h : nullptr;
}
int kref_gpu_transform(int inverse, uint64_t t, const uint8_t* src, uint64_t n, uint8_t* dst, uint64_t cap, uint64_t* out_n) {
    KREF_TRY
    void* h = open_handle(1u << 20);
    if (!h) { g_err = "knz_open failed"; return 5; }
    auto [tr, err] = kz_transform::NewGPUTransform(go_unsafe::Pointer(h), go::Uint64(go::U(t)));
    int rc = 0;
    if (err != nullptr) rc = fail(err, 2);
    else {
        go::Slice<go::Byte> in = copy_in(src, n);
        go::Slice<go::Byte> outb = go::Slice<go::Byte>::make((int64_t)cap, (int64_t)cap);
        auto [rd, wr, ferr] = inverse ? tr->Inverse(in, outb) : tr->Forward(in, outb);
        if (ferr != nullptr) { fail(ferr, -1); rc = inverse ? 3 : -1; }
        else { if (wr.v) memcpy(dst, outb.p, wr.v); *out_n = wr.v; }
    }
    ::knz_close(h);
    return rc;
    KREF_CATCH
}
int kref_gpu_entropy_encode(uint32_t type, const uint8_t* src, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_bits) {
    KREF_TRY
    void* h = open_handle(1u << 20);
    if (!h) { g_err = "knz_open failed"; return 5; }
    MemStream ms;
    auto [obs, err0] = kz_bitstream::NewDefaultOutputBitStream(&ms, go::Uint(go::U(16384)));
    auto [ee, err] = kz_entropy::NewGPUEntropyEncoder(go_unsafe::Pointer(h), obs, go::Uint32(go::U(type)));
    int rc = 0;
    if (err != nullptr) rc = fail(err, 2);
    else {
        auto [wr, werr] = ee->Write(copy_in(src, n));
        if (werr != nullptr) rc = fail(werr, 3);
        else {
            ee->Dispose();
            obs->Close();
            if (ms.data.size() > cap) { g_err = "output buffer too small"; rc = 4; }
            else { memcpy(out, ms.data.data(), ms.data.size()); *out_bits = obs->Written().v; }
        }
    }
    ::knz_close(h);
    return rc;
    KREF_CATCH
}
#endif

uint32_t kref_xxhash32(const uint8_t* d, uint64_t n, uint32_t seed) {
    go::ArenaScope arena_;
    auto [h, err] = kz_hash::NewXXHash32(go::Uint32(go::U(seed)));
    return h->Hash(view(d, n)).v;
}
uint64_t kref_xxhash64(const uint8_t* d, uint64_t n, uint64_t seed) {
    go::ArenaScope arena_;
    auto [h, err] = kz_hash::NewXXHash64(go::Uint64(go::U(seed)));
    return h->Hash(view(d, n)).v;
}
uint32_t kref_magic_type(const uint8_t* d, uint64_t n) { go::ArenaScope arena_; return kz_internal::GetMagicType(view(d, n)).v; }
// internal.ComputeFirstOrderEntropy1024 over the order-0 histogram (the -s / skipBlocks test of io/CompressedStream.go:778-800)
int kref_entropy1024(const uint8_t* d, uint64_t n) {
    go::ArenaScope arena_;
    go::Array<go::Int, 256> freqs{};
    kz_internal::ComputeHistogram(view(d, n), go::slice(freqs, go::none, go::none), true, false);
    return (int)kz_internal::ComputeFirstOrderEntropy1024(go::Int(go::U(n)), go::slice(freqs, go::none, go::none)).v;
}
// transform.GetType(name) / entropy.GetType(name): the ids the stream header carries
uint64_t kref_transform_type(const char* name) {
    go::ArenaScope arena_;
    try { auto [t, err] = kz_transform::GetType(go::String(name)); return err != nullptr ? ~0ull : (uint64_t)t.v; } catch (...) { return ~0ull; }
}
uint32_t kref_entropy_type(const char* name) {
    go::ArenaScope arena_;
    try { auto [t, err] = kz_entropy::GetType(go::String(name)); return err != nullptr ? ~0u : (uint32_t)t.v; } catch (...) { return ~0u; }
}

}  // extern "C"
