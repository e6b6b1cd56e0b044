#!/usr/bin/env python3
"""The host-pointer boundary (knz_encode_blocks / knz_decode_blocks) with the caller's buffers in pageable memory against pinned memory, and the raw copy rates
of the box beside it (what bounds the hook on the bandwidth-shaped configuration). usage: hook_pinned.py [config=huffman] [lanes=1,3]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np, torch, bench, bench_corpus
K = bench.load_pkg(); K.build_library()
from kanzi_go_amd import api as A
cfg = sys.argv[1] if len(sys.argv) > 1 else "huffman"
lanes_list = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,3").split(",")]
transform, entropy, bs, _i, _c = bench.CONFIGS[cfg]
data = bench_corpus.s_silesia()
n = len(data)
dev = torch.device("cuda", 0)
out = {"config": f"-t {transform} -e {entropy} -b {bs >> 20}m", "bytes": n}

def rate(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return n * reps / 1e9 / (time.perf_counter() - t0)

d = torch.empty(n, dtype=torch.uint8, device=dev)
hp = torch.from_numpy(np.array(data, copy=True))
hpin = torch.empty(n, dtype=torch.uint8).pin_memory(); hpin.copy_(hp)
out["raw_copy_GBps"] = {"h2d_pageable": round(rate(lambda: d.copy_(hp, non_blocking=True)), 1), "h2d_pinned": round(rate(lambda: d.copy_(hpin, non_blocking=True)), 1),
                        "d2h_pageable": round(rate(lambda: hp.copy_(d, non_blocking=True)), 1), "d2h_pinned": round(rate(lambda: hpin.copy_(d, non_blocking=True)), 1)}
nb = (n + bs - 1) // bs

def buffers(pinned, count, size):
    if pinned:
        t = torch.empty(count * size, dtype=torch.uint8).pin_memory()
        return [t[i * size:(i + 1) * size].numpy() for i in range(count)], t
    return [np.zeros(size, dtype=np.uint8) for _ in range(count)], None

rows = []
for pinned in (False, True):
    for lanes in lanes_list:
        c = K.Codec(transform, entropy, bs, devices=[0] * lanes)
        cap = int(c.L.knz_max_encoded_len(c.cfg.transform, bs)) * 2 + 262144
        srcs, k1 = buffers(pinned, nb, bs)
        for i in range(nb):
            blk = data[i * bs:(i + 1) * bs]
            srcs[i][:len(blk)] = blk
        outs, k2 = buffers(pinned, nb, cap)
        arr = (A._Block * nb)()
        for i in range(nb):
            arr[i].src = srcs[i].ctypes.data; arr[i].src_len = min(bs, n - i * bs); arr[i].dst = outs[i].ctypes.data; arr[i].dst_cap = cap
        te = td = 1e9
        for r in range(4):
            t0 = time.perf_counter(); c._chk(c.L.knz_encode_blocks(c.h, arr, nb)); dt = time.perf_counter() - t0
            if r: te = min(te, dt)
        pays, k3 = buffers(pinned, nb, cap)
        backs, k4 = buffers(pinned, nb, bs + max(512, bs >> 4))
        arr2 = (A._Block * nb)()
        for i in range(nb):
            m = (arr[i].out_bits + 7) // 8
            pays[i][:m] = outs[i][:m]
            arr2[i].src = pays[i].ctypes.data; arr2[i].src_len = m; arr2[i].dst = backs[i].ctypes.data; arr2[i].dst_cap = len(backs[i])
        for r in range(4):
            t0 = time.perf_counter(); c._chk(c.L.knz_decode_blocks(c.h, arr2, nb)); dt = time.perf_counter() - t0
            if r: td = min(td, dt)
        ok = all(bytes(backs[i][: arr2[i].out_bits]) == data[i * bs:(i + 1) * bs].tobytes() for i in range(nb))
        rows.append({"host_memory": "pinned" if pinned else "pageable", "lanes": lanes, "encode_MBps": round(n / 1e6 / te, 1), "decode_MBps": round(n / 1e6 / td, 1),
                     "round_trip_MBps": round(n / 1e6 / (te + td), 1), "ok": ok})
        c.close()
out["hook"] = rows
print(json.dumps(out))
