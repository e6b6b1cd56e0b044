#!/usr/bin/env python3
"""Per-block time of the LZ forward chain on S-silesia (4 MiB blocks through knz_transform_forward): which blocks set the batch time."""
import json, os, struct, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, bench_corpus, knz  # noqa
K = knz.package(); K.build_library()
data = bench_corpus.s_silesia(); bs = 4 << 20
c = K.Codec("NONE", "NONE", bs); t = K.ByteTransform(c, "LZ")
out = []
step = int(os.environ.get("STEP", "3"))
for bi in range(0, (len(data) + bs - 1) // bs, step):
    blk = data[bi * bs:(bi + 1) * bs].tobytes()
    t.forward(blk[:65536])
    t0 = time.perf_counter(); f = t.forward(blk); dt = time.perf_counter() - t0
    rec = {"block": bi, "ms": round(dt * 1e3, 1), "applied": f is not None}
    if f is not None:
        litEnd, tk, m = struct.unpack("<III", f[:12]); rec.update({"out": len(f), "literals": litEnd - 13, "tokens": tk})
    out.append(rec)
print(json.dumps(out))
