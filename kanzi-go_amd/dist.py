"""Multi-GPU sharding of a stream: static contiguous block ranges per rank, one variable-size gather of
the compressed segments to rank 0 (RCCL over xGMI with backend "nccl"; gloo in the CPU tests), bit-granular assembly
on rank 0 (`knz_dev_assemble`). No collective sits on the per-block data path."""
import torch
import torch.distributed as dist


def block_range(nblocks, rank, world):
    """Blocks [lo, hi) owned by `rank`: contiguous ranges so that the gather is one message per rank."""
    per = (nblocks + world - 1) // world
    return min(rank * per, nblocks), min((rank + 1) * per, nblocks)


class _PendingGather:
    """A gather of the rank segments that is still in flight (RCCL runs it on its own stream): `finish()` waits for it and,
    on rank 0, assembles the stream. Whatever the caller launches in between (e.g. decoding its own segment, which needs
    nothing from the other ranks) overlaps with the transfer over xGMI."""

    def __init__(self, codec, work, bufs, bits, nbits, total_size, out, stream):
        self.codec, self.work, self.bufs, self.bits, self.nbits = codec, work, bufs, bits, nbits
        self.total_size, self.out, self.stream = total_size, out, stream

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.bufs is None:
            return None, self.nbits
        nbytes = self.codec.dev_assemble(self.total_size, [b.data_ptr() for b in self.bufs], self.bits, self.out.data_ptr(),
                                         self.out.numel(), stream=self.stream)
        return nbytes, self.nbits


def gather_segments(seg, seg_bits, group=None, async_op=False):
    """seg: uint8 tensor holding this rank's bit string (zero padded), seg_bits: its bit count.
    Returns (work handle or None, list of per-rank tensors on rank 0 / None elsewhere, list of bit counts)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = seg.device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([seg_bits], dtype=torch.int64, device=dev), group=group)
    bits = [int(s.item()) for s in sizes]
    maxb = ((max(bits) + 7) // 8 + 8 + 15) & ~15
    if maxb > seg.numel():
        raise ValueError("segment buffers must be sized identically on every rank")
    bufs = [torch.empty(maxb, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
    work = dist.gather(seg[:maxb], bufs, dst=0, group=group, async_op=async_op)
    return (work if async_op else None), bufs, bits


def sharded_compress_begin(codec, d_src, n_local, seg, total_size, out, stream=0, group=None):
    """Every rank encodes its own blocks into `seg` and the gather to rank 0 is started; returns (pending, nbits):
    `pending.finish()` -> (byte length of the assembled stream in `out` on rank 0 / None elsewhere, nbits)."""
    nbits = codec.dev_compress_blocks(d_src.data_ptr(), n_local, seg.data_ptr(), seg.numel(), stream=stream) if n_local else 0
    work, bufs, bits = gather_segments(seg, nbits, group, async_op=True)
    return _PendingGather(codec, work, bufs, bits, nbits, total_size, out, stream), nbits


def sharded_compress(codec, d_src, n_local, seg, total_size, out, stream=0, group=None):
    """Every rank encodes its own blocks into `seg`; rank 0 returns the byte length of the assembled stream in `out`."""
    pending, _ = sharded_compress_begin(codec, d_src, n_local, seg, total_size, out, stream=stream, group=group)
    return pending.finish()
