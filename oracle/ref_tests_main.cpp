// Runner of the reference's OWN unit tests, translated (tools/go2cpp) -- TEST INFRASTRUCTURE.
//
// `make -C oracle _ref_tests` translates kanzi-go's *_test.go files together with the sources they test (entropy/Entropy_test.go,
// transform/Transforms_test.go, transform/BWT_test.go, transform/EXECodec_test.go, bitstream/DefaultBitstream_test.go,
// io/CompressedStream_test.go) where they lie under /root/reference and links them with this file: every `func TestXxx(t *testing.T)` is
// registered under "package.TestXxx" (runtime/go_rt.hpp go_testing). `make -C oracle _ref_gpu_tests` does the same with the cgo shim of go/,
// the hooks of go/testhooks and tools/go2cpp/apply_test_patch.py, linked against libknz_gpu.so: the same tests then build their codecs,
// transforms and streams on the device.
//
//   knz_ref_tests --list              names, one per line
//   knz_ref_tests [NAME ...]          runs the named tests (all when none is given): "NAME ok" / "NAME FAIL" + the test's log; exit 1 on a failure
//   KREF_TEST_SEED=n                  seed of the math/rand shim (the tests draw their random inputs from it)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifdef KREF_WITH_CGO
#include "cgo_shim.hpp"
#include "kanzi_ref_gpu_tests.gen.hpp"
#else
#include "kanzi_ref_tests.gen.hpp"
#endif

int main(int argc, char** argv) {
    if (const char* s = std::getenv("KREF_TEST_SEED")) go_rand::state() = std::strtoull(s, nullptr, 0);
    std::vector<std::string> want;
    for (int i = 1; i < argc; i++) {
        if (!std::strcmp(argv[i], "--list")) {
            for (auto& e : go_testing::registry()) std::printf("%s\n", e.first.c_str());
            return 0;
        }
        want.push_back(argv[i]);
    }
    int bad = 0, ran = 0;
    for (auto& e : go_testing::registry()) {
        bool take = want.empty();
        for (auto& w : want) take = take || w == e.first;
        if (!take) continue;
        std::string log;
        int rc;
        {
            go::ArenaScope arena;                      // everything a test allocates goes with it
            rc = go_testing::run(e.first, log);
        }
        std::printf("%s %s\n", e.first.c_str(), rc == 0 ? "ok" : "FAIL");
        if (rc != 0) { std::printf("%s", log.c_str()); bad++; }
        std::fflush(stdout);
        ran++;
    }
#ifdef KREF_WITH_CGO
    std::printf("device objects: entropy %lld transform %lld streams %lld\n", (long long)kz_entropy::GpuTestObjects().v, (long long)kz_transform::GpuTestObjects().v,
                (long long)kz_io::GpuTestStreams().v);
#endif
    if (ran == 0) { std::fprintf(stderr, "no such test\n"); return 2; }
    return bad ? 1 : 0;
}
