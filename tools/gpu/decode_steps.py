#!/usr/bin/env python3
"""Wall time of every single decode of the default configuration's stream (one process, N decodes in a row): how the chains' time is distributed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bench_corpus, knz
K = knz.package(); K.build_library()
data = bench_corpus.s_silesia(); n = len(data)
dev = torch.device("cuda", 0)
src = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
dst = torch.zeros(n + n // 2, dtype=torch.uint8, device=dev); back = torch.zeros(n, dtype=torch.uint8, device=dev)
c = K.Codec("BWT+RANK+ZRLT", "ANS1", 8 << 20)
nb = c.dev_compress(src.data_ptr(), n, dst.data_ptr(), dst.numel()); torch.cuda.synchronize()
ts = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 24):
    torch.cuda.synchronize(); t0 = time.time()
    m = c.dev_decompress(dst.data_ptr(), nb, back.data_ptr(), n)
    torch.cuda.synchronize(); ts.append((time.time() - t0) * 1e3)
assert m == n and bool(torch.equal(src, back))
print("decode ms:", " ".join(f"{t:.1f}" for t in ts))
print(f"min {min(ts):.1f} median {sorted(ts)[len(ts) // 2]:.1f} max {max(ts):.1f}")
