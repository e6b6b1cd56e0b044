"""Development aid: builds libknz_gpu_prof.so with -DKNZ_PROFILE_PHASES (cycle counters around kernel phases) and prints the
per-phase totals of one decode of bench config 2. Not part of the product or the tests."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import knz as KK
import bench_corpus
K = KK.package()
out = os.path.join(ROOT, "gpurun_out", "libknz_gpu_prof.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DKNZ_PROFILE_PHASES",
                       "-Wno-unused-variable", "-Wno-unused-value", "-o", out, os.path.join(ROOT, "kanzi-go_amd", "csrc", "knz_gpu.hip")])
torch.zeros(1, device="cuda")
L = K.load_library(out)
size = int(sys.argv[1]) if len(sys.argv) > 1 else bench_corpus.SILESIA_SIZE
data = torch.from_numpy(bench_corpus.s_silesia(size)).cuda()
n = data.numel()
c = K.Codec("NONE", "HUFFMAN", 4 << 20, lib=out, device=0)
cap = 2 * n + (1 << 22)
dst = torch.empty(cap, dtype=torch.uint8, device="cuda")
back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
nb = c.dev_compress(data.data_ptr(), n, dst.data_ptr(), cap)
for it in range(2):
    L.knz_debug_prof(None, 1)
    nd = c.dev_decompress(dst.data_ptr(), nb, back.data_ptr(), n + 64)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
L.knz_debug_prof(buf, 0)
names = {0: "dec stage+parse", 1: "dec table", 2: "dec sync passes", 3: "dec write pass", 4: "dec copy out", 5: "dec total",
         8: "walk fill+issue", 9: "walk window copy", 10: "walk parse", 11: "walk ring store",
         16: "EG rounds (both)", 17: "parse alphabet", 18: "parse EG rounds", 19: "parse scan+end", 20: "parse lengths", 21: "parse varints",
         22: "parses", 23: "sum count", 24: "chunks with a write pass"}
chunks = (n + 16383) // 16384
for i, nm in names.items():
    print(f"{nm:20s} total {buf[i]:>14d} ticks   per chunk {buf[i] / chunks:10.1f}")
print("roundtrip ok", bool(torch.equal(back[:n], data)), "timing", c.last_timing())
