#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer batch hook (knz_encode_blocks / knz_decode_blocks: what the cgo shim of
Writer.processBlock / Reader.processBlock calls): blocks in pageable host memory in, block-local streams in host memory out.
Reported in BASELINE.md next to the device-resident rate of bench.py; never bench.py's `value`."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import bench_corpus  # noqa: E402


def main():
    import torch
    torch.zeros(1, device="cuda:0")
    K = bench.load_pkg()
    data = bench_corpus.s_silesia()
    bs = 4 << 20
    blocks = [data[i:i + bs] for i in range(0, len(data), bs)]
    c = K.Codec("NONE", "HUFFMAN", bs, device=0)
    import numpy as np
    from kanzi_go_amd import api as A

    def batch(srcs, cap):
        arr = (A._Block * len(srcs))()
        keep = []
        for i, b in enumerate(srcs):
            a = np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else np.ascontiguousarray(b)
            o = np.zeros(cap, dtype=np.uint8)
            keep.append((a, o))
            arr[i].src = a.ctypes.data; arr[i].src_len = len(a); arr[i].dst = o.ctypes.data; arr[i].dst_cap = cap
        return arr, keep

    def timed(fn, arr, n):                       # only the C call is timed (second call: the workspace exists)
        c._chk(fn(c.h, arr, n))
        t0 = time.perf_counter(); c._chk(fn(c.h, arr, n)); return time.perf_counter() - t0

    arr, keep = batch(blocks, 2 * bs + 262144)
    te = timed(c.L.knz_encode_blocks, arr, len(blocks))
    payloads = [keep[i][1][: (arr[i].out_bits + 7) // 8].copy() for i in range(len(blocks))]
    arr2, keep2 = batch(payloads, bs + max(512, bs >> 4))
    td = timed(c.L.knz_decode_blocks, arr2, len(blocks))
    assert all(bytes(keep2[i][1][: arr2[i].out_bits]) == bytes(blocks[i]) for i in range(len(blocks)))
    out = {"what": "knz_encode_blocks / knz_decode_blocks (C call only), 51 x 4 MiB blocks in pageable host memory, -t NONE -e HUFFMAN",
           "encode_MBps": round(len(data) / 1e6 / te, 1), "decode_MBps": round(len(data) / 1e6 / td, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
