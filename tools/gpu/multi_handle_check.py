#!/usr/bin/env python3
"""K host threads x one handle: knz_encode_blocks / knz_decode_blocks at the same time, every result compared with the input (and the encoder's bytes with
handle 0's). usage: multi_handle_check.py K [rounds]   (GPU_MAX_HW_QUEUES=32 in the environment makes the streams of the handles truly concurrent)"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench, bench_corpus
K = bench.load_pkg()
from kanzi_go_amd import api as A
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bs, per = 8 << 20, 26
data = bench_corpus.s_silesia()
blocks = [np.ascontiguousarray(data[i * bs:(i + 1) * bs]) for i in range(25)] + [np.ascontiguousarray(data[:bs])]
codecs = [K.Codec("BWT+RANK+ZRLT", "ANS1", bs, 0, 0) for _ in range(k)]
cap = int(codecs[0].L.knz_max_encoded_len(codecs[0].cfg.transform, bs)) * 2 + 262144
bad = 0
for r in range(rounds):
    encs, outs = [], []
    for t in range(k):
        arr = (A._Block * per)(); o = [np.zeros(cap, dtype=np.uint8) for _ in range(per)]
        for i, a in enumerate(blocks):
            arr[i].src = a.ctypes.data; arr[i].src_len = len(a); arr[i].dst = o[i].ctypes.data; arr[i].dst_cap = cap
        encs.append(arr); outs.append(o)
    def run(fn, arrs):
        res = [0] * k
        def work(t):
            res[t] = getattr(codecs[t].L, fn)(codecs[t].h, arrs[t], per)
        ths = [threading.Thread(target=work, args=(t,)) for t in range(k)]
        t0 = time.perf_counter()
        [th.start() for th in ths]; [th.join() for th in ths]
        return res, time.perf_counter() - t0
    res, dt = run("knz_encode_blocks", encs)
    print(f"round {r}: encode rc {res} {k * per * bs / 1e6 / dt:.0f} MB/s")
    decs, douts = [], []
    for t in range(k):
        arr = (A._Block * per)(); pays = [outs[t][i][: (encs[t][i].out_bits + 7) // 8].copy() for i in range(per)]
        o = [np.zeros(bs + (bs >> 4), dtype=np.uint8) for _ in range(per)]
        for i in range(per):
            arr[i].src = pays[i].ctypes.data; arr[i].src_len = len(pays[i]); arr[i].dst = o[i].ctypes.data; arr[i].dst_cap = len(o[i])
            if t and not np.array_equal(pays[i], decs[0][1][i]):
                print("  encode of handle", t, "block", i, "differs from handle 0"); bad += 1
        decs.append((arr, pays)); douts.append(o)
    res, dt = run("knz_decode_blocks", [d[0] for d in decs])
    print(f"round {r}: decode rc {res} {k * per * bs / 1e6 / dt:.0f} MB/s  piped blocks {[c.last_counter(6) for c in codecs]}")
    for t in range(k):
        if res[t]:
            print("  handle", t, "error:", codecs[t].L.knz_last_error(codecs[t].h).decode(), "status", [decs[t][0][i].status for i in range(per)]); bad += 1
            continue
        for i in range(per):
            if decs[t][0][i].out_bits != len(blocks[i]) or not np.array_equal(douts[t][i][: len(blocks[i])], blocks[i]):
                print("  handle", t, "block", i, "decodes to something else"); bad += 1
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
fr, tot = ctypes.c_size_t(), ctypes.c_size_t()
hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot))
print(f"device memory in use with {k} handles open: {(tot.value - fr.value) / 2**30:.1f} GiB of {tot.value / 2**30:.0f}")
print("BAD" if bad else "OK", bad)
