"""RCCL under the multi-GPU code on the ONE GPU a test box has: backend "nccl" (= RCCL on ROCm) initialised with device_id at world_size 1, the
8-byte all_gather of the segment sizes, a grouped isend / irecv (to itself: ncclGroupStart .. ncclSend / ncclRecv .. ncclGroupEnd), then the sharded
pipeline of kanzi-go_amd/dist.py (encode the rank's blocks, gather, bit-granular assembly on rank 0, decode of the rank's own segment) against the
oracle. The 8-GPU run of the driver is then not RCCL's first contact with this code. The N > 1 logic itself is covered by tests/test_dist_gloo.py."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
import knz, parity_cases as P, oracle_lib as O
K = knz.package()
from kanzi_go_amd import dist as kd
assert torch.cuda.is_available()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
# the collective and the grouped point-to-point the gather is made of, on RCCL
sizes = [torch.zeros(1, dtype=torch.int64, device=dev)]
dist.all_gather(sizes, torch.tensor([12345], dtype=torch.int64, device=dev))
assert int(sizes[0].item()) == 12345
a = torch.arange(4096, dtype=torch.uint8, device=dev)
b = torch.zeros(4096, dtype=torch.uint8, device=dev)
for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, b, 0), dist.P2POp(dist.isend, a, 0)]):
    w.wait()
torch.cuda.synchronize()
assert torch.equal(a, b), "grouped send / recv over RCCL"
# the sharded pipeline (world_size 1: rank 0 owns every block and gathers nothing), both the blocking and the overlapped form
for transform, entropy, bs, n in (("BWT+RANK+ZRLT", "ANS1", 1 << 16, 5 * (1 << 16) + 4321), ("LZ", "ANS0", 1 << 16, 3 * (1 << 16) + 99), ("NONE", "HUFFMAN", 1 << 16, 70000)):
    data = P.corpus(n, 5)
    codec = K.Codec(transform, entropy, bs, device=0)
    src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
    nblocks = (n + bs - 1) // bs
    seg = torch.zeros(2 * nblocks * bs + (1 << 18), dtype=torch.uint8, device=dev)
    out = torch.zeros(2 * n + (1 << 18), dtype=torch.uint8, device=dev)
    pending, nbits = kd.sharded_compress_begin(codec, src, n, seg, n, out)
    back = torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    assert codec.dev_decompress_blocks(seg.data_ptr(), nbits, back.data_ptr(), back.numel()) == n      # (overlaps the gather on N > 1)
    nbytes, _ = pending.finish()
    assert out[:nbytes].cpu().numpy().tobytes() == O.compress(data, transform, entropy, bs), "assembled stream differs from the oracle"
    assert back[:n].cpu().numpy().tobytes() == data
    codec.close()
dist.barrier()
dist.destroy_process_group()
print("rccl ok")
'''


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_world_size_one(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(31500 + (os.getpid() % 2000))
    p = subprocess.run([sys.executable, str(script), HERE, port], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=580)
    assert p.returncode == 0, p.stdout[-3000:]
    assert "rccl ok" in p.stdout


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_distributed_code_path_on_rccl():
    """bench.py --force-dist: the N > 1 code path of the bench itself (process group on RCCL, sharded encode, gather, bit-granular assembly, per-rank
    decode, the max-over-ranks collectives) at world size 1, on the full configs[3] job: the JSON contract and the assembled stream against the oracle.
    (Config 5 at its stated 10^9 bytes is not in the suite for time: one 32 MiB block and a ragged one are, test_config5_block_size_32m_fpaq; the full
    size is a builder-run line, profiles/r04_final_config_fpaq_bench.json.)"""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, MASTER_PORT=str(29600 + (os.getpid() % 300)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--steps", "1", "--warmup", "1", "--no-pmc", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=850, cwd=root, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 1 and out["scaling"] == "strong" and out["dtype"] == "u8"
    assert "configs[3]" in out["config"]["workload"] and out["config"]["blocks"] == 26
    assert out["roundtrip_ok"] is True and out["bit_exact_vs_oracle"] is True
    assert out["roofline"]["kernel"] is not None and out["roofline"]["encode_MBps"] > 0 and out["roofline"]["decode_MBps"] > 0
    assert out["value"] > 100                                   # (MB/s: the default job's round trip is ~340 on one MI355X)
