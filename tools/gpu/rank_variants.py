#!/usr/bin/env python3
"""A/B of the inverse RANK chain variants (rank_inv.hip, KNZ_RANK_VARIANT) on real pipeline data: 8 MiB blocks of
S-silesia through the device's BWT and RANK, then knz_transform_inverse(RANK) timed per variant (one block = one chain:
the batch of 26 runs the same chains side by side, so a block's time is the stage's time). Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (brings up the HIP runtime)
import bench_corpus  # noqa: E402
import knz  # noqa: E402

K = knz.package()
K.build_library()
data = bench_corpus.s_silesia()
bs = 8 << 20
out = {"block_bytes": bs, "variants": {"0": "round-1 kernel", "3": "select step, SALU arithmetic, per-lane threshold",
                                        "5": "select step on the vector ALU (default)"}, "blocks": []}
c = K.Codec("NONE", "NONE", bs)
bwt, rank = K.ByteTransform(c, "BWT"), K.ByteTransform(c, "RANK")
nblk = (len(data) + bs - 1) // bs
ab = {0, 7, 18}                                            # blocks timed with every variant; the others with the default only
for bi in range(nblk):
    off = bi * bs
    blk = data[off:off + bs].tobytes()
    b = bwt.forward(blk)
    r = rank.forward(b)
    a = np.frombuffer(r, dtype=np.uint8)
    rec = {"block": bi, "zero_frac": round(float((a == 0).mean()), 4), "ge64_frac": round(float((a >= 64).mean()), 6),
           "zero_word_frac": round(float((a[: len(a) // 4 * 4].reshape(-1, 4).max(axis=1) == 0).mean()), 4), "ms": {}}
    for v in (os.environ.get("VARIANTS", "0,3,5").split(",") if bi in ab else ["5"]):
        os.environ["KNZ_RANK_VARIANT"] = v
        best = 1e9
        for _ in range(2 if bi in ab else 1):
            t0 = time.perf_counter()
            back = rank.inverse(r, len(b) + 512)
            best = min(best, time.perf_counter() - t0)
        assert back == b, v
        rec["ms"][v] = round(best * 1e3, 1)
    out["blocks"].append(rec)
os.environ.pop("KNZ_RANK_VARIANT", None)
out["slowest_block_ms_default"] = max(r["ms"]["5"] for r in out["blocks"])
print(json.dumps(out))
