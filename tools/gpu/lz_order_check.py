#!/usr/bin/env python3
"""Cross-check for the cross-lane ordering the segment-parallel LZ parse relies on (wave.h: wave_order_lanes): the hole bits a wave sets
with atomicOr from 64 lanes are read back by the same wave with relaxed agent-scope loads, with no fence in between. The argument why that is ordered
on gfx950 is in wave.h; this script is the empirical half: the executable-like blocks of S-silesia (the ones with the most jumped-over positions) are
parsed ITER times by the segment-parallel form under uneven load (a second stream keeps the chip busy with copies) and every result is compared with the
first-form one-wave parse (KNZ_LZ_CHAIN), which has no such hand-over. usage: lz_order_check.py [ITER=1000]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bench_corpus, knz
ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = knz.package(); K.build_library()
bs = 4 << 20
data = np.concatenate([bench_corpus._segment("exe", 6 * bs, 2), bench_corpus._segment("img16", bs, 3), bench_corpus._segment("records", bs, 4)])
n = len(data)
dev = torch.device("cuda", 0)
src = torch.from_numpy(data).to(dev)
dst = torch.zeros(n + n // 2, dtype=torch.uint8, device=dev)
os.environ["KNZ_LZ_CHAIN"] = "1"
c = K.Codec("LZ", "NONE", bs)
nb = c.dev_compress(src.data_ptr(), n, dst.data_ptr(), dst.numel())
ref = dst[:nb].clone()
c.close()
del os.environ["KNZ_LZ_CHAIN"]
c = K.Codec("LZ", "NONE", bs)
noise = torch.cuda.Stream()
a = torch.empty(64 << 20, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
bad = 0
for it in range(ITER):
    if it % 3:                                          # uneven load: copies on another stream during two of three iterations
        with torch.cuda.stream(noise):
            for _ in range(1 + it % 5):
                b.copy_(a, non_blocking=True)
    m = c.dev_compress(src.data_ptr(), n, dst.data_ptr(), dst.numel())
    if m != nb or not torch.equal(dst[:m], ref):
        bad += 1
        print("iteration", it, "differs from the one-wave parse", flush=True)
torch.cuda.synchronize()
print(f"{ITER} iterations, {n} bytes in {n // bs} blocks each, rounds {c.last_counter(5)}, one-wave blocks {c.last_counter(4)}: {bad} mismatches")
sys.exit(1 if bad else 0)
