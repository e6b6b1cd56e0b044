#!/usr/bin/env python3
"""Per-launch times of the segment-parallel LZ parse (one line per round) on S-silesia, -t LZ -e ANS0 -b 4m."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bench_corpus, knz
K = knz.package(); K.build_library()
data = bench_corpus.s_silesia()
n = len(data)
dev = torch.device("cuda", 0)
src = torch.from_numpy(data).to(dev)
dst = torch.zeros(n + n // 2, dtype=torch.uint8, device=dev)
c = K.Codec("LZ", "ANS0", 4 << 20)
for it in range(2):
    nb = c.dev_compress(src.data_ptr(), n, dst.data_ptr(), dst.numel())
    torch.cuda.synchronize()
print("stream bytes", nb, "rounds", c.last_counter(5), "one-wave blocks", c.last_counter(4))
for name, ms in c.last_kernel_times():
    print("%-40s %.3f ms" % (name.split("(")[0][-40:], ms))
