"""bench.py's N > 1 step (strong scaling = the default: one fixed job split over the ranks; weak: one corpus copy per rank, one stream; gather to rank 0, assembly, per-rank decode)
exercised without GPUs: launched exactly as the driver launches it (torch.distributed.run, one process per rank), with
KNZ_BENCH_EMU=1 = CPU tensors + kernels on tests/emu + gloo. Checks the JSON contract and that the assembled stream is the
oracle's stream of the whole N-copy input."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, extra):
    import knz
    knz.emu_library()                                    # build once, before the ranks race for it
    env = dict(os.environ, KNZ_BENCH_EMU="1")
    port = str(23000 + (os.getpid() % 4000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"] + extra
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(1000)
@pytest.mark.parametrize("nproc", [2, 3])
def test_bench_ranks_weak_scaling(nproc):
    size, bs = 5 * 65536 + 4321, 65536
    out = _run(nproc, ["--config", "huffman", "--scaling", "weak", "--size", str(size), "--block-size", str(bs)])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == nproc and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True
    assert out["config"]["blocks"] == (nproc * size + bs - 1) // bs
    assert out["roundtrip_ok"] is True
    assert out["bit_exact_vs_oracle"] is True
    assert "EMULATOR" in out["data"]


@pytest.mark.timeout(1000)
@pytest.mark.parametrize("nproc,nblk", [(2, 5), (3, 2)])
def test_bench_ranks_strong_scaling_default_config(nproc, nblk):
    """--scaling strong: ONE fixed job of BASELINE configs[3]'s pipeline (BWT+RANK+ZRLT / ANS1) split over the ranks by
    contiguous block ranges; (3 ranks, 2 blocks) leaves a rank without any block."""
    bs = 16384
    size = (nblk - 1) * bs + 4321
    out = _run(nproc, ["--size", str(size), "--block-size", str(bs)])           # (strong is the default: the figure BASELINE.json's metric names)
    assert out["scaling"] == "strong" and out["n_gpus"] == nproc
    # the weak figure rides along as a second timed region, as scalars inside `roofline` (where the driver's parser keeps them)
    assert out["roofline"]["weak_value_MBps"] > 0 and out["roofline"]["weak_blocks"] == (nproc * size + bs - 1) // bs
    assert out["weak_scaling"]["roundtrip_ok_rank0"] is True
    for key in ("encode_MBps", "decode_MBps", "encode_ms", "decode_ms"):
        assert out["roofline"][key] > 0, key
    assert "configs[3]" in out["config"]["workload"] and "BWT+RANK+ZRLT" in out["config"]["workload"]
    assert out["config"]["blocks"] == nblk
    assert out["roundtrip_ok"] is True and out["bit_exact_vs_oracle"] is True
    assert out["roofline"]["kernel"] is not None and out["roofline"]["avg_launch_ms"] >= 0


@pytest.mark.timeout(1000)
def test_bench_ranks_explicit_weak_on_the_default_config():
    """--scaling weak on the default pipeline: one corpus copy per rank in one stream, no second region."""
    bs = 16384
    size = 2 * bs + 999
    out = _run(2, ["--scaling", "weak", "--size", str(size), "--block-size", str(bs)])
    assert out["scaling"] == "weak" and out["n_gpus"] == 2
    assert "configs[3]" in out["config"]["workload"] and "2 copies" in out["config"]["workload"]
    assert out["config"]["blocks"] == (2 * size + bs - 1) // bs
    assert out["roundtrip_ok"] is True and out["bit_exact_vs_oracle"] is True
    assert "weak_value_MBps" not in out["roofline"]
