"""Multi-GPU sharding of a stream: static contiguous block ranges per rank, one variable-size gather of
the compressed segments to rank 0 (RCCL over xGMI with backend "nccl"; gloo in the CPU tests), bit-granular assembly
on rank 0 (`knz_dev_assemble`). No collective sits on the per-block data path."""
import torch
import torch.distributed as dist


def block_range(nblocks, rank, world):
    """Blocks [lo, hi) owned by `rank`: contiguous ranges so that the gather is one message per rank."""
    per = (nblocks + world - 1) // world
    return min(rank * per, nblocks), min((rank + 1) * per, nblocks)


def gather_segments(seg, seg_bits, group=None):
    """seg: uint8 tensor holding this rank's bit string (zero padded), seg_bits: its bit count.
    Returns (list of per-rank tensors, list of bit counts) on rank 0, (None, bits) elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = seg.device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([seg_bits], dtype=torch.int64, device=dev), group=group)
    bits = [int(s.item()) for s in sizes]
    maxb = ((max(bits) + 7) // 8 + 8 + 15) & ~15
    if maxb > seg.numel():
        raise ValueError("segment buffers must be sized identically on every rank")
    if rank == 0:
        bufs = [torch.empty(maxb, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(seg[:maxb], bufs, dst=0, group=group)
        return bufs, bits
    dist.gather(seg[:maxb], None, dst=0, group=group)
    return None, bits


def sharded_compress(codec, d_src, n_local, seg, total_size, out, stream=0, group=None):
    """Every rank encodes its own blocks into `seg`; rank 0 returns the byte length of the assembled stream in `out`."""
    nbits = codec.dev_compress_blocks(d_src.data_ptr(), n_local, seg.data_ptr(), seg.numel(), stream=stream) if n_local else 0
    bufs, bits = gather_segments(seg, nbits, group)
    if bufs is None:
        return None, nbits
    nbytes = codec.dev_assemble(total_size, [b.data_ptr() for b in bufs], bits, out.data_ptr(), out.numel(), stream=stream)
    return nbytes, nbits
