"""Parity tests proper: the gfx950 library on a real MI355X, called through the C ABI, against the CPU oracle."""
import os

import numpy as np
import pytest

import parity_cases as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    P.K.build_library()
    return P.GpuBackend()


@pytest.mark.parametrize("etype", ["HUFFMAN", "NONE"])
def test_entropy_objects_bit_exact(be, etype):
    P.check_entropy_encode(be, etype)


@pytest.mark.parametrize("cfg", [
    ("NONE", "HUFFMAN", 1 << 16, 300000), ("NONE", "HUFFMAN", 1 << 16, (1 << 16) + 5), ("NONE", "HUFFMAN", 1024, 1000),
    ("NONE", "HUFFMAN", 1024, 10), ("NONE", "HUFFMAN", 4096, 4096 * 3 + 15), ("NONE", "NONE", 1 << 16, 200003),
    ("NONE", "HUFFMAN", 1 << 20, 5 * (1 << 20) + 17), ("NONE", "HUFFMAN", 4 << 20, 3 * (4 << 20) + 12345),
])
def test_stream_bit_exact(be, cfg):
    P.check_stream(be, *cfg)


def test_block_batch_hook(be):
    P.check_block_batch(be, "NONE", "HUFFMAN", 1 << 16, 3, 12345)
    P.check_block_batch(be, "NONE", "HUFFMAN", 1 << 16, 1, 9)
    P.check_block_batch(be, "NONE", "HUFFMAN", 4 << 20, 2, 1 << 20)


@pytest.mark.parametrize("ranks", [1, 2, 8])
def test_multi_gpu_assemble_single_device(be, ranks):
    P.check_assemble(be, "HUFFMAN", 1 << 16, 5 * (1 << 16) + 777, ranks)


def test_full_size_config2_properties(be):
    """BASELINE.json configs[1] at full size (S-silesia, 211,957,760 B, -b 4m): size-independent properties —
    encode -> decode round trip on the device, and a checksum-of-chunks comparison with the oracle stream."""
    import hashlib
    import bench_corpus
    import oracle_lib as O
    import torch
    data = bench_corpus.s_silesia()
    n, bs = len(data), 4 << 20
    c = P.K.Codec("NONE", "HUFFMAN", bs, lib=be.lib)
    d_src = torch.from_numpy(data).to(be.dev)
    cap = n + n // 2
    d_dst = torch.zeros(cap, dtype=torch.uint8, device=be.dev)
    nb = c.dev_compress(d_src.data_ptr(), n, d_dst.data_ptr(), cap)
    d_back = torch.zeros(n + 4096, dtype=torch.uint8, device=be.dev)
    assert c.dev_decompress(d_dst.data_ptr(), nb, d_back.data_ptr(), n + 4096) == n
    assert torch.equal(d_back[:n], d_src)
    got = d_dst[:nb].cpu().numpy().tobytes()
    exp = O.compress(data, "NONE", "HUFFMAN", bs, 0, jobs=os.cpu_count() or 1)
    assert len(got) == len(exp)
    assert hashlib.sha256(got).digest() == hashlib.sha256(exp).digest()
    c.close()


def test_stress_inputs(be):
    import bench_corpus
    for gen in (bench_corpus.s_rand, bench_corpus.s_ramp):
        data = gen(3 * (1 << 20) + 77).tobytes()
        be_c = P.K.Codec("NONE", "HUFFMAN", 1 << 20, lib=be.lib)
        import oracle_lib as O
        src, ks = be.to_dev(data)
        cap = len(data) * 2
        dst, kd = be.empty(cap)
        nb = be_c.dev_compress(src, len(data), dst, cap)
        assert be.to_host(kd, nb) == O.compress(data, "NONE", "HUFFMAN", 1 << 20)
        be_c.close()
    data = bytes(5 * (1 << 20))
    P.check_stream(be, "NONE", "HUFFMAN", 1 << 20, len(data))
