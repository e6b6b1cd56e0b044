"""N > 1 path on CPU: world_size 2, gloo, kernels on the emulator (tests/emu). Each rank encodes its contiguous block
range, segments are gathered to rank 0 and assembled bit-granularly; the result must equal the oracle's stream and
every rank must decode its own segment back."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
import knz, parity_cases as P, oracle_lib as O
K = knz.package()
from kanzi_go_amd import dist as kd
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
rank, world = dist.get_rank(), dist.get_world_size()
entropy, bs, n = sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
transform = sys.argv[8]
data = P.corpus(n, 5)
if transform.startswith('TEXT'):
    import text_corpus
    data = text_corpus.make_text(n, seed=5, utf8=0.02)
nblocks = (n + bs - 1) // bs
lo_b, hi_b = kd.block_range(nblocks, rank, world)
lo, hi = lo_b * bs, min(hi_b * bs, n)
part = data[lo:hi]
codec = K.Codec(transform, entropy, bs, lib=knz.emu_library())
per = kd.max_blocks_per_rank(nblocks, world)
cap = 2 * per * bs + (1 << 18)
raw = torch.zeros(len(part) + 64, dtype=torch.uint8)
off = (-raw.data_ptr()) % 16
src = raw[off:off + max(len(part), 1)]
if part:
    src[: len(part)] = torch.from_numpy(np.frombuffer(part, dtype=np.uint8).copy())
seg = torch.zeros(cap + 16, dtype=torch.uint8)
seg = seg[(-seg.data_ptr()) % 16:][:cap]
out = torch.zeros(2 * n + (1 << 18) + 16, dtype=torch.uint8)
out = out[(-out.data_ptr()) % 16:]
nbytes, nbits = kd.sharded_compress(codec, src, len(part), seg, n, out)
if rank == 0:
    got = out[:nbytes].numpy().tobytes()
    assert got == O.compress(data, transform, entropy, bs), "assembled stream differs from the oracle"
back = torch.zeros(len(part) + 64, dtype=torch.uint8)
if part:
    assert codec.dev_decompress_blocks(seg.data_ptr(), nbits, back.data_ptr(), back.numel()) == len(part)
    assert back[: len(part)].numpy().tobytes() == part
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


@pytest.mark.parametrize("cfg", [("HUFFMAN", 1 << 16, 5 * (1 << 16) + 777, 2, "NONE"), ("ANS0", 1 << 16, 2 * (1 << 16) + 5, 2, "NONE"),
                                 ("HUFFMAN", 1 << 16, 1000, 2, "NONE"),
                                 ("ANS1", 1 << 14, 4 * (1 << 14) + 4321, 2, "BWT+RANK+ZRLT"),      # configs[3] pipeline, uneven last rank (3 + 2 blocks, ragged tail)
                                 ("ANS1", 1 << 14, 2 * (1 << 14) + 99, 3, "BWT+RANK+ZRLT"),        # 3 blocks over 3 ranks, the last one 99 bytes
                                 ("ANS0", 1 << 14, 4 * (1 << 14) + 321, 2, "TEXT+UTF+BWT+RANK+ZRLT"),    # the -l 5 sequence on text, 3 + 2 blocks
                                 # the bench's block -> rank map: 26 blocks (25 full + a short one, like S-silesia at -b 8m) over 4 ranks = 7,7,6,6
                                 ("ANS1", 1 << 12, 25 * (1 << 12) + 1095, 4, "BWT+RANK+ZRLT"),
                                 ("ANS0", 1 << 12, 25 * (1 << 12) + 1095, 4, "LZ")])                      # the pipeline that scales (configs[2]), same map
def test_two_ranks_gloo(cfg, tmp_path):
    entropy, bs, n, world, transform = cfg
    import knz
    knz.emu_library()                                    # build once, before the ranks race for it
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + (os.getpid() % 2000))
    procs = [subprocess.Popen([sys.executable, str(script), HERE, port, str(r), str(world), entropy, str(bs), str(n), transform],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o
