"""Helper of tests/test_go_shim_emu.py, run in a child process with the HIP emulator build of the library preloaded in place of libknz_gpu.so:
the reference's Writer / Reader through the cgo shim of go/ (oracle/_ref/libknz_ref_gpu.so) against the kernels on the emulator. TEST INFRASTRUCTURE."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity_cases as P          # noqa: E402
import ref_lib as R               # noqa: E402
import test_go_shim_gpu as T      # noqa: E402


def main():
    L = C.CDLL(T.SO)
    u8p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    L.kref_last_error.restype = C.c_char_p
    L.kref_gpu_compress.argtypes = [u8p, C.c_uint64, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, C.c_int, u8p, C.c_uint64, u64p]
    L.kref_gpu_decompress.argtypes = [u8p, C.c_uint64, C.c_uint32, u8p, C.c_uint64, u64p]
    cases = 0
    for transform, entropy, bs, ck in (("BWT+RANK+ZRLT", "ANS1", 1 << 14, 64), ("LZ", "HUFFMAN", 1 << 14, 32), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0", 1 << 14, 0),
                                       ("NONE", "NONE", 1 << 14, 0)):
        for n in (0, 1, 40, 3 * bs + 77, 2 * bs):
            data = P.corpus(n, seed=7 + n % 13)
            R.record_events(5)
            stream = R.compress(data, transform, entropy, bs, ck, jobs=1)
            want_w = R.event_log()
            assert R.decompress(stream, n + 64, jobs=1) == data
            want_r = R.event_log()
            R.record_events(-1)
            for jobs, depth in ((1, 0), (16, 0), (2, 100)):       # depth: Writer / Reader.EnableGPUDepth (a batch depth of its own; 0 = `jobs` blocks per batch)
                L.kref_gpu_depth(depth)
                R.record_events(5, L)
                assert T.gpu_compress(L, data, transform, entropy, bs, ck, jobs=jobs) == stream, (transform, entropy, n, jobs, depth, "stream")
                got_w = R.event_log(L)
                assert T.gpu_decompress(L, stream, n + 64, jobs=jobs) == data, (transform, entropy, n, jobs, "decode")
                got_r = R.event_log(L)
                R.record_events(-1, L)
                assert got_w == want_w, (transform, entropy, n, jobs, "Writer events", [(a, b) for a, b in zip(got_w, want_w) if a != b][:3], len(got_w), len(want_w))
                assert got_r == want_r, (transform, entropy, n, jobs, "Reader events", [(a, b) for a, b in zip(got_r, want_r) if a != b][:3], len(got_r), len(want_r))
                cases += 1
            L.kref_gpu_depth(0)
    # several lanes behind the one handle of a Writer / Reader (EnableGPUDevices), block ranges, damaged streams
    # (kept small here: the MI355X suite runs K = 1, 2, 3, 8 over six pipelines, tests/test_go_shim_gpu.py)
    cases += T.check_lanes(L, [("BWT+RANK+ZRLT", "ANS1", 1 << 13, 64)], lambda bs: (1, 5 * bs + 321), (2, 3), ((16, 0), (3, 128)))
    cases += T.check_lanes(L, [("LZ", "HUFFMAN", 1 << 14, 0)], lambda bs: (21 * bs,), (8,), ((3, 128),))
    cases += T.check_ranges_and_damage(L, [("NONE", "HUFFMAN", 1 << 13, 32)], lanes_list=(0, 2))
    print(f"go shim on the emulator: {cases} cases ok")


if __name__ == "__main__":
    main()
