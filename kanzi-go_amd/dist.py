"""Multi-GPU sharding of a stream: static contiguous block ranges per rank, one variable-size gather of
the compressed segments to rank 0 (RCCL over xGMI with backend "nccl"; gloo in the CPU tests), bit-granular assembly
on rank 0 (`knz_dev_assemble`). No collective sits on the per-block data path."""
import torch
import torch.distributed as dist


def blocks_per_rank(nblocks, world):
    """Static scatter: the first nblocks % world ranks own one block more (26 blocks over 8 GPUs: 4,4,3,3,3,3,3,3)."""
    q, rem = divmod(nblocks, world)
    return [q + (1 if r < rem else 0) for r in range(world)]


def max_blocks_per_rank(nblocks, world):
    return (nblocks + world - 1) // world


def block_range(nblocks, rank, world):
    """Blocks [lo, hi) owned by `rank`: contiguous ranges so that the gather is one message per rank."""
    q, rem = divmod(nblocks, world)
    lo = rank * q + min(rank, rem)
    return lo, lo + q + (1 if rank < rem else 0)


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


class _Works:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


def gather_segments(seg, seg_bits, group=None, async_op=False):
    """seg: uint8 tensor holding this rank's bit string (zero padded), seg_bits: its bit count.
    Variable-size gather to rank 0: the bit counts are all-gathered (8 bytes per rank), then every rank sends
    exactly its own bytes and rank 0 posts one receive per rank: one grouped send/recv (ncclGroupStart .. ncclSend/ncclRecv ..
    ncclGroupEnd under backend "nccl" = RCCL), nothing is padded to the largest segment.
    Returns (work handle or None, list of per-rank tensors on rank 0 / None elsewhere, list of bit counts)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = seg.device
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([seg_bits], dtype=torch.int64, device=dev), group=group)
    bits = [int(s.item()) for s in sizes]
    nbytes = [((b + 7) // 8 + 3) & ~3 for b in bits]                     # whole 32-bit words: the assembly reads words
    if nbytes[rank] > seg.numel():
        raise ValueError("segment buffer smaller than the segment")
    ops, bufs = [], None
    if rank == 0:
        bufs = [seg[: nbytes[0]]] + [torch.empty(max(nbytes[r], 4), dtype=torch.uint8, device=dev) for r in range(1, world)]
        ops = [dist.P2POp(dist.irecv, bufs[r][: nbytes[r]], r, group) for r in range(1, world) if nbytes[r]]
    elif nbytes[rank]:
        ops = [dist.P2POp(dist.isend, seg[: nbytes[rank]], 0, group)]
    works = dist.batch_isend_irecv(ops) if ops else []
    work = _Works(works)
    if not async_op:
        work.wait()
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
