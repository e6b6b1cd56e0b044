#!/usr/bin/env python3
"""One block of (nearly) the largest size the format allows (default 2^30 - 4096 bytes: a BWT block of exactly 2^30 bytes grows by its 33-byte header past
the 2^30 the reference's decoder accepts as a block length, CompressedStream.go:1893, mirrored in knz_walk_block_header) through the device: BWT forward, then
the device's own inverse BWT (an independent implementation) must give the input back, for the BWT alone and for the bench pipeline. No oracle: its
DivSufSort restatement needs minutes for a block of this size. usage: big_block_check.py [log2 of the size, default 30]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bench_corpus, knz
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = (1 << lg) - 4096
K = knz.package(); K.build_library()
data = bench_corpus.s_enwik(n)
dev = torch.device("cuda", 0)
src = torch.from_numpy(data).to(dev)
dst = torch.zeros(n + n // 4 + (1 << 20), dtype=torch.uint8, device=dev)
back = torch.zeros(n + 4096, dtype=torch.uint8, device=dev)
for transform, entropy in (("BWT", "NONE"), ("BWT+RANK+ZRLT", "ANS0")):
    c = K.Codec(transform, entropy, n)
    t0 = time.perf_counter()
    nb = c.dev_compress(src.data_ptr(), n, dst.data_ptr(), dst.numel())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    nd = c.dev_decompress(dst.data_ptr(), nb, back.data_ptr(), back.numel())
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ok = nd == n and bool(torch.equal(back[:n], src))
    print(f"{transform}/{entropy}: one block of 2^{lg} - 4096 bytes -> {nb} bytes, encode {t1 - t0:.2f} s, decode {t2 - t1:.2f} s, round trip {'ok' if ok else 'FAILED'}", flush=True)
    c.close()
    if not ok:
        sys.exit(1)
