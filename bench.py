#!/usr/bin/env python3
"""bench.py — encode+decode throughput of the MI355X-native Kanzi hot path (BASELINE.json metric).

Workload (N=1): BASELINE.json configs[1] = `-t NONE -e HUFFMAN -b 4m` on S-silesia (211,957,760 synthetic bytes
shaped like silesia.tar, bench_corpus.py). One step = compress the whole stream on the device (bit-exact .knz) and
decompress it back, inputs resident in HBM. value = uncompressed MB (10^6 B) per second of a whole round trip;
encode-only and decode-only rates are reported next to it.

N>1 (weak scaling: per-GPU work is fixed): the stream is N copies of S-silesia back to back (N x 211,957,760 B, one .knz
stream); its blocks are sharded statically (contiguous ranges, ~51 blocks per rank) over the ranks; every rank
encodes/decodes its own blocks, the compressed segments are gathered to rank 0 over RCCL and assembled bit-granularly
there; the gather stays in flight while each rank decodes its own segment. value = all bytes of all ranks / step time.

KNZ_BENCH_EMU=1 is a TEST HARNESS switch (tests/test_bench_ranks.py): the same control flow on CPU tensors with the kernels
compiled against tests/emu and gloo instead of RCCL, so that the N>1 step can be exercised without GPUs. It is not a
product path and its JSON says so.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

CONFIGS = {
    # name: (transform, entropy, block size, BASELINE.json config index)
    "huffman": ("NONE", "HUFFMAN", 4 << 20, 1),
    "ans0": ("NONE", "ANS0", 4 << 20, 2),       # entropy half of configs[2] (LZ front end: see DESIGN.md)
    "lz": ("LZ", "ANS0", 4 << 20, 2),
    "bwt": ("BWT+RANK+ZRLT", "ANS1", 8 << 20, 3),
    "fpaq": ("BWT+RANK+ZRLT", "FPAQ", 32 << 20, 4),   # configs[4] codec on S-silesia-shaped data (enwik9-sized input: --size 1000000000)
}


def load_pkg():
    spec = importlib.util.spec_from_file_location("kanzi_go_amd", os.path.join(ROOT, "kanzi-go_amd", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(ROOT, "kanzi-go_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["kanzi_go_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(data, transform, entropy, bs, budget_s=20.0):
    """The oracle (C++ restatement of the kanzi-go CPU path, kind 'port') on the host cores, bounded sample."""
    import oracle_lib as O
    cores = os.cpu_count() or 1
    sample = data
    # probe on 32 MiB, then size the sample to ~budget_s of CPU work
    probe = data[: min(len(data), 32 << 20)]
    t0 = time.perf_counter()
    c = O.compress(probe, transform, entropy, bs, 0, jobs=cores)
    O.decompress(c, len(probe) + 64, jobs=cores)
    dt = time.perf_counter() - t0
    rate = len(probe) / dt
    n = int(min(len(data), max(len(probe), rate * budget_s)))
    n = max(bs, (n // bs) * bs)
    sample = data[:n]
    t0 = time.perf_counter()
    c = O.compress(sample, transform, entropy, bs, 0, jobs=cores)
    t1 = time.perf_counter()
    back = O.decompress(c, len(sample) + 64, jobs=cores)
    t2 = time.perf_counter()
    assert back == sample.tobytes()
    return {
        "value": round(len(sample) / 1e6 / (t2 - t0), 2), "unit": "MB/s", "cores": cores, "kind": "port",
        "encode_MBps": round(len(sample) / 1e6 / (t1 - t0), 2), "decode_MBps": round(len(sample) / 1e6 / (t2 - t1), 2),
        "sample": f"first {len(sample)} bytes of S-silesia, {transform}/{entropy} -b {bs}, round trip, one block per thread",
        "note": "C++ restatement of the kanzi-go CPU path (no Go toolchain in the image)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="huffman", choices=sorted(CONFIGS))
    ap.add_argument("--size", type=int, default=0, help="override the per-GPU corpus size (debug)")
    ap.add_argument("--block-size", type=int, default=0, help="override the block size (debug)")
    ap.add_argument("--transform", default="", help="override the transform sequence of the config, e.g. LZP or BWT+SRT+ZRLT (debug)")
    ap.add_argument("--entropy", default="", help="override the entropy codec of the config (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import bench_corpus

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    emu = os.environ.get("KNZ_BENCH_EMU") == "1"          # test harness only, see the module docstring
    if emu:
        import knz
        dev = torch.device("cpu")
        sync = lambda: None
        if world > 1:
            dist.init_process_group("gloo")
        K = knz.package()
        lib = knz.emu_library()
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        sync = torch.cuda.synchronize
        if world > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", device_id=dev)
        K = load_pkg()
        K.build_library()
        lib = None
    from kanzi_go_amd import dist as kd
    transform, entropy, bs, cfg_idx = CONFIGS[args.config]
    transform, entropy = args.transform or transform, args.entropy or entropy
    bs = args.block_size or bs

    base_size = args.size or bench_corpus.SILESIA_SIZE
    base = bench_corpus.s_silesia(base_size)
    size = base_size * world                                 # weak scaling: one copy of the corpus per GPU, ONE stream
    nblocks = (size + bs - 1) // bs
    lo_b, hi_b = kd.block_range(nblocks, rank, world)
    per = (nblocks + world - 1) // world
    lo, hi = lo_b * bs, min(hi_b * bs, size)

    def tiled(a, b):                                         # bytes [a, b) of the corpus repeated back to back
        parts = []
        while a < b:
            o = a % base_size
            take = min(b - a, base_size - o)
            parts.append(base[o:o + take])
            a += take
        return np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)

    my = tiled(lo, hi)
    n_my = len(my)

    def dev_zeros(n):                                        # 16-byte aligned (the emulator's "device" memory is host memory)
        t = torch.zeros(n + 16, dtype=torch.uint8, device=dev)
        return t[(-t.data_ptr()) % 16:][:n]

    codec = K.Codec(transform, entropy, bs, device=local_rank, lib=lib)
    d_src = dev_zeros(max(n_my, 16))
    if n_my:
        d_src[:n_my] = torch.from_numpy(np.ascontiguousarray(my)).to(dev)
    cap = (per * bs + (per * bs) // 2 + (1 << 20) + 15) & ~15   # same on every rank (gather uses equal-sized buffers)
    d_seg = dev_zeros(cap)
    d_back = dev_zeros(n_my + 4096)
    d_stream = dev_zeros(size + size // 2 + (1 << 20)) if rank == 0 and world > 1 else None
    stream = 0 if emu else torch.cuda.current_stream().cuda_stream

    stage = {"enc_transform": 0.0, "enc_entropy": 0.0, "enc_layout": 0.0, "enc_gather": 0.0, "dec_walk": 0.0, "dec_entropy": 0.0, "dec_transform": 0.0}
    t_enc = t_dec = 0.0
    result = {}

    def one_step(timed):
        nonlocal t_enc, t_dec
        t0 = time.perf_counter()
        if world == 1:
            nb = codec.dev_compress(d_src.data_ptr(), n_my, d_seg.data_ptr(), cap, header_input_size=size, stream=stream)
            result["stream_bytes"] = nb
        else:
            # every rank encodes its blocks; the gather of the segments to rank 0 over RCCL/xGMI is started and stays in flight
            # while the rank decodes its own segment (which needs nothing from the others); then rank 0 assembles the stream
            pending, nbits = kd.sharded_compress_begin(codec, d_src, n_my, d_seg, size, d_stream if rank == 0 else d_seg, stream=stream)
            result["seg_bits"] = nbits
        tm = codec.last_timing()
        if world == 1:
            sync()
        t1 = time.perf_counter()
        if world == 1:
            nd = codec.dev_decompress(d_seg.data_ptr(), result["stream_bytes"], d_back.data_ptr(), d_back.numel(), stream=stream)
        else:
            nd = codec.dev_decompress_blocks(d_seg.data_ptr(), result["seg_bits"], d_back.data_ptr(), d_back.numel(), stream=stream) if n_my else 0
        td = codec.last_timing()
        sync()
        t2 = time.perf_counter()
        if world > 1:
            nb, _ = pending.finish()
            sync()
            t3 = time.perf_counter()
            t0 -= (t3 - t2)                        # the assembly belongs to the encode side of the step
            if rank == 0:
                result["stream_bytes"] = nb
        assert nd == n_my, (nd, n_my)
        if timed:
            t_enc += t1 - t0
            t_dec += t2 - t1
            stage["enc_transform"] += tm[0]; stage["enc_entropy"] += tm[1]; stage["enc_layout"] += tm[2]; stage["enc_gather"] += tm[3]
            stage["dec_walk"] += td[0]; stage["dec_entropy"] += td[1]; stage["dec_transform"] += td[2]

    for _ in range(args.warmup):
        one_step(False)
    if world > 1:
        dist.barrier()
    sync()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        one_step(True)
    sync()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if world > 1:
        tt = torch.tensor([elapsed, t_enc, t_dec], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, t_enc, t_dec = [float(x) for x in tt.tolist()]

    ok_roundtrip = bool(torch.equal(d_back[:n_my], d_src[:n_my])) if n_my else True
    if world > 1:                                            # every rank's decode must have come back right
        okt = torch.tensor([1 if ok_roundtrip else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok_roundtrip = bool(int(okt.item()))

    if rank == 0:
        K_ = max(args.steps, 1)
        ms = elapsed / K_ * 1e3
        C_bytes = result["stream_bytes"]
        out = {
            "metric": "encode+decode MB/s (round trip of the whole stream, uncompressed 10^6 B per second)",
            "value": round(size / 1e6 / (elapsed / K_), 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic" if not emu else "synthetic (EMULATOR TEST HARNESS on CPU: not a measurement)",
            "config": {"workload": f"BASELINE.json configs[{cfg_idx}]: -t {transform} -e {entropy} -b {bs >> 20}m on "
                                   + (f"{world} x " if world > 1 else "") + f"S-silesia ({base_size} B per GPU, one stream of {size} B, bench_corpus.py)",
                       "blocks": nblocks, "block_size": bs,
                       "parallelism": f"contiguous block ranges over {world} GPU(s), segments gathered to rank 0"},
            "encode_MBps": round(size / 1e6 / (t_enc / K_), 2), "decode_MBps": round(size / 1e6 / (t_dec / K_), 2),
            "compressed_bytes": int(C_bytes), "roundtrip_ok": ok_roundtrip,
        }
        # roofline of the dominant kernel, timed with HIP events on the launch stream inside the library
        per_launch = {k: v / K_ for k, v in stage.items()}
        n_local, c_local = n_my, (result.get("seg_bits", C_bytes * 8) + 7) // 8 if world > 1 else C_bytes
        kern = {
            {"HUFFMAN": "knz_huf_hist+lengths+encode_kernels", "ANS0": "knz_ans0_stats+encode_kernels", "ANS1": "knz_ans1_hist+stats+merge+encode_kernels", "FPAQ": "knz_fpaq_encode_kernel", "NONE": "knz_raw_units_kernel"}[entropy]: (per_launch["enc_entropy"], n_local + c_local),
            {"HUFFMAN": "knz_huf_walk_decode_kernel", "ANS0": "knz_ans0_walk_decode_kernel", "ANS1": "knz_ans1_dec_tables+decode_kernels", "FPAQ": "knz_fpaq_decode_kernel", "NONE": "knz_huf_decode_kernel (raw copy)"}[entropy]: (per_launch["dec_entropy"], n_local + c_local),
            ("knz_dec_block_headers_kernel" if entropy in ("HUFFMAN", "ANS0") else "knz_dec_walk_blocks_kernel"): (per_launch["dec_walk"], c_local),
            "forward transform stage kernels (" + transform + ")": (per_launch["enc_transform"], 2 * n_local),
            "inverse transform stage kernels (" + transform + ")": (per_launch["dec_transform"], 2 * n_local),
        }
        dom = max(kern, key=lambda k: kern[k][0])
        dur_ms, alg = kern[dom]
        ach = alg / 1e9 / (dur_ms / 1e3) if dur_ms > 0 else 0.0
        traffic = None
        try:   # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/, see its _how)
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_config2_v4.json")))
            if args.config == "huffman" and world == 1 and not args.size and dom in pm["kernels"]:
                traffic = pm["kernels"][dom]["hbm_bytes_corrected"]
        except Exception:
            traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
                           "algorithmic_bytes_per_launch": int(alg), "avg_launch_ms": round(dur_ms, 4),
                           "all_stage_ms": {k: round(v, 4) for k, v in per_launch.items()}}
        if not args.no_verify:
            import oracle_lib as O
            exp = O.compress(tiled(0, size), transform, entropy, bs, 0, jobs=os.cpu_count() or 1)
            got = (d_seg if world == 1 else d_stream)[:C_bytes].cpu().numpy().tobytes()
            out["bit_exact_vs_oracle"] = bool(got == exp)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(base, transform, entropy, bs)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
