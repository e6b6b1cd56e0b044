#!/usr/bin/env python3
"""Batches of very many small blocks (default 70,000 blocks of 1 KiB; arguments: blocks, block size, "transform/entropy,...") through the bench pipeline and the BWT alone: stream == oracle, round trip.
(The suffix sort takes them as ONE group since round 4; rounds 1-3 split batches into groups of 1023 blocks.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bench_corpus, knz, oracle_lib as O
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = nblk * bs - 77
K = knz.package(); K.build_library()
data = bench_corpus.s_silesia(max(n, 1 << 20))[:n]
dev = torch.device("cuda", 0)
src = torch.from_numpy(np.ascontiguousarray(data)).to(dev)
dst = torch.zeros(2 * n + 64 * nblk + (1 << 20), dtype=torch.uint8, device=dev)
back = torch.zeros(n + 4096, dtype=torch.uint8, device=dev)
pairs = (("BWT", "NONE"), ("BWT+RANK+ZRLT", "ANS0"), ("BWT+RANK+ZRLT", "HUFFMAN"))
if len(sys.argv) > 3:                                                     # e.g. "BWT+RANK+ZRLT/ANS1,LZ/ANS0"
    pairs = tuple(tuple(p.split("/")) for p in sys.argv[3].split(","))
for transform, entropy in pairs:
    c = K.Codec(transform, entropy, bs)
    t0 = time.perf_counter()
    nb = c.dev_compress(src.data_ptr(), n, dst.data_ptr(), dst.numel())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    exp = O.compress(data, transform, entropy, bs, 0, jobs=os.cpu_count() or 1)
    same = dst[:nb].cpu().numpy().tobytes() == exp
    nd = c.dev_decompress(dst.data_ptr(), nb, back.data_ptr(), back.numel())
    ok = nd == n and bool(torch.equal(back[:n], src))
    print(f"{transform}/{entropy}: {nblk} blocks of {bs} bytes -> {nb} bytes in {t1 - t0:.2f} s, stream == oracle: {same}, round trip: {ok}", flush=True)
    c.close()
    if not (same and ok):
        sys.exit(1)
