#!/usr/bin/env python3
"""FPAQ chains alone on the device: -t NONE -e FPAQ -b 4m on the first 32 MiB of S-silesia (8 blocks = 8 chains side by side), and on its
BWT+RANK+ZRLT form (what configs[4] feeds the coder). Prints MB/s per chain (block bytes / wall time of the batch) and checks the round trip."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bench_corpus, knz
K = knz.package(); K.build_library()
bs = 4 << 20
data = bench_corpus.s_silesia()[: 8 * bs]
dev = torch.device("cuda", 0)
for transform in ("NONE", "BWT+RANK+ZRLT"):
    c = K.Codec(transform, "FPAQ", bs)
    n = len(data)
    src = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
    dst = torch.zeros(n + n // 2, dtype=torch.uint8, device=dev)
    back = torch.zeros(n, dtype=torch.uint8, device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        nb = c.dev_compress(src.data_ptr(), n, dst.data_ptr(), dst.numel())
        torch.cuda.synchronize(); t1 = time.time()
        m = c.dev_decompress(dst.data_ptr(), nb, back.data_ptr(), n)
        torch.cuda.synchronize(); t2 = time.time()
    ok = m == n and bool(torch.equal(src, back))
    print(f"-t {transform} -e FPAQ -b 4m, 8 blocks: {nb} bytes, encode {t1 - t0:.3f} s = {bs / 1e6 / (t1 - t0):.2f} MB/s per chain, "
          f"decode {t2 - t1:.3f} s = {bs / 1e6 / (t2 - t1):.2f} MB/s per chain, round trip {'ok' if ok else 'BROKEN'}", flush=True)
    c.close()
