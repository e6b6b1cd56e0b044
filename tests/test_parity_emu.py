"""HIP kernels vs the CPU oracle, run on the HIP execution-model emulator (tests/emu, test infrastructure):
the same kernel sources that hipcc compiles for gfx950 are executed thread-by-thread on the CPU so that indexing
and bit-layout mistakes are caught in the build container. The real-GPU run of the same cases is
tests/test_parity_gpu.py."""
import os

import pytest

import parity_cases as P


@pytest.fixture(scope="module")
def be():
    return P.EmuBackend()


@pytest.mark.parametrize("sched", ["fwd", "rev"])
@pytest.mark.parametrize("etype", ["HUFFMAN", "NONE", "ANS0", "ANS1", "FPAQ"])
def test_entropy_objects_bit_exact(be, etype, sched, monkeypatch):
    monkeypatch.setenv("KNZ_EMU_SCHED", sched)   # thread order inside a workgroup: catches missing barriers
    P.check_entropy_encode(be, etype)


@pytest.mark.parametrize("cfg", [
    ("NONE", "HUFFMAN", 1 << 16, 300000), ("NONE", "HUFFMAN", 1 << 16, 1 << 16), ("NONE", "HUFFMAN", 1 << 16, (1 << 16) + 5),
    ("NONE", "HUFFMAN", 1024, 1000), ("NONE", "HUFFMAN", 1024, 10), ("NONE", "HUFFMAN", 1 << 20, 70000),
    ("NONE", "HUFFMAN", 4096, 4096 * 3 + 15), ("NONE", "NONE", 1 << 16, 200003), ("NONE", "HUFFMAN", 1 << 20, (1 << 20) + 17),
    ("NONE", "ANS0", 1 << 16, 300000), ("NONE", "ANS0", 1024, 1000), ("NONE", "ANS0", 1024, 10), ("NONE", "ANS0", 1 << 16, (1 << 16) + 33),
    ("NONE", "ANS0", 1 << 20, (1 << 20) + 17), ("NONE", "ANS0", 4096, 4096 * 2 + 3),
    ("NONE", "ANS1", 1 << 16, 300000), ("NONE", "ANS1", 1024, 1000), ("NONE", "ANS1", 1024, 10),
    ("BWT", "HUFFMAN", 1 << 16, 200000), ("BWT+RANK+ZRLT", "ANS0", 1 << 14, 40000), ("BWT+RANK+ZRLT", "ANS1", 1 << 14, 50000),
    ("BWT+MTFT+ZRLT", "ANS0", 1 << 14, 20003), ("RANK", "HUFFMAN", 1 << 14, 20000), ("ZRLT", "NONE", 1 << 16, 150000),
    ("BWT+RANK+ZRLT", "ANS1", 1024, 1000), ("BWT+RANK+ZRLT", "ANS1", 1024, 12),
    ("NONE", "FPAQ", 1 << 16, 300000), ("NONE", "FPAQ", 1024, 1000), ("NONE", "FPAQ", 1024, 10), ("BWT+RANK+ZRLT", "FPAQ", 1 << 13, 20000),
    ("LZ", "ANS0", 1 << 16, 300000), ("LZ", "HUFFMAN", 1 << 18, 300000), ("LZX", "HUFFMAN", 1 << 16, 150000), ("LZ", "ANS0", 1024, 1000), ("LZ", "ANS0", 1024, 20),
    ("BWT+ZRLT", "NONE", 1024, 1024 * 1030 + 5),      # > 1023 blocks: the suffix sort runs in groups
    ("BWT+SRT+ZRLT", "ANS0", 1 << 14, 40000), ("LZP", "HUFFMAN", 1 << 16, 200000), ("SRT", "NONE", 1024, 1000), ("LZP+SRT", "ANS0", 1 << 15, 70000),
])
def test_stream_bit_exact(be, cfg):
    P.check_stream(be, *cfg)


def test_stream_random_schedule(be, monkeypatch):
    monkeypatch.setenv("KNZ_EMU_SCHED", "rand")
    P.check_stream(be, "NONE", "HUFFMAN", 1 << 16, 150001, seed=8)


def test_block_batch_hook(be):
    P.check_block_batch(be, "NONE", "HUFFMAN", 1 << 16, 3, 12345)
    P.check_block_batch(be, "NONE", "HUFFMAN", 1 << 16, 1, 9)      # copy block (<= 15 bytes)
    P.check_block_batch(be, "NONE", "NONE", 4096, 2, 4096)
    P.check_block_batch(be, "NONE", "ANS0", 1 << 16, 3, 33)


@pytest.mark.parametrize("lanes", [1, 2, 3, 8])
def test_multi_device_batch_hook(be, lanes):
    # row e' of the round-5 verdict: several devices behind the C ABI the Go host binds (knz_open_devices), here as logical lanes on the emulator
    P.check_multi_device_batch(be, "NONE", "HUFFMAN", 1 << 14, 7, 1234, lanes)
    P.check_multi_device_batch(be, "NONE", "ANS0", 1 << 14, 2, 9, lanes, checksum_bits=32)        # fewer blocks than lanes, a copy block last
    P.check_multi_device_batch(be, "BWT+RANK+ZRLT", "ANS1", 1 << 12, 5, 777, lanes)


@pytest.mark.parametrize("ranks", [1, 2, 3, 8])
def test_multi_gpu_assemble(be, ranks):
    P.check_assemble(be, "HUFFMAN", 1 << 16, 5 * (1 << 16) + 777, ranks)
    P.check_assemble(be, "ANS0", 1 << 16, 3 * (1 << 16) + 5, ranks)


@pytest.mark.parametrize("tname", ["ZRLT", "RANK", "MTFT", "BWT", "LZ", "LZX", "SRT", "LZP", "UTF"])
def test_transform_objects_bit_exact(be, tname):
    # the register-resident SBRT list uses ~12 cross-lane operations per byte: keep the emulated inputs small
    P.check_transform(be, tname, max_len=4096 if tname in ("RANK", "MTFT") else (32768 if tname == "SRT" else 1 << 30))


def test_block_batch_hook_transforms(be):
    P.check_block_batch(be, "BWT+RANK+ZRLT", "ANS1", 1 << 13, 3, 4321)
    P.check_block_batch(be, "BWT+RANK+ZRLT", "ANS0", 1 << 13, 2, 9)


def test_huffman_encoder_scratch_form(be, monkeypatch):
    # default = units encoded at their final bit positions (sizes pass, layout scans, encoder); KNZ_HUF_SCRATCH = the round-1 form (units to
    # scratch slots, knz_gather_kernel), which -s streams still take
    monkeypatch.setenv("KNZ_HUF_SCRATCH", "1")
    P.check_entropy_encode(be, "HUFFMAN")
    P.check_stream(be, "NONE", "HUFFMAN", 1 << 16, 300000)
    P.check_stream(be, "NONE", "HUFFMAN", 4096, 4096 * 3 + 15)


def test_huffman_decoder_paths(be):
    P.check_huffman_shapes(be)


def test_block_checksums(be):
    P.check_checksums(be)


def test_ans1_table_decoder(be):
    P.check_ans1_table_decoder(be)


def test_huffman_split_walk(be):
    P.check_huffman_split_walk(be)


def test_mtft_segments(be):
    P.check_mtft_segments(be)


def test_reference_test_inputs_both_directions(be):
    P.check_reference_inputs(be)


def test_device_vs_ref_directly(be):
    """the reference's own code (oracle/_ref = kanzi-go's sources translated by tools/go2cpp) as the checker, nothing hand-written in between"""
    import ref_lib as R
    if not R.available():
        pytest.skip("oracle/_ref is not built")
    P.check_device_vs_ref(be, quick=True)


def test_batch_split_when_workspace_is_refused(be, monkeypatch):
    P.check_alloc_split(be, monkeypatch)


def test_short_block_inside_stream(be):
    P.check_short_inner_block(be)


def test_bwt_inverse_list_ranking(be, monkeypatch):
    P.check_bwt_list_ranking(be, monkeypatch, max_len=2100)
    monkeypatch.setenv("KNZ_BWT_RANK_MIN", "256")
    P.check_corrupt_streams(be, trials=2)


def test_bwt_suffix_sort_forms(be, monkeypatch):
    P.check_bwt_sort_forms(be, monkeypatch)


def test_bwt_suffix_sort_wide_keys(be, monkeypatch):
    P.check_bwt_sort_wide_keys(be, monkeypatch)


def test_bwt_suffix_sort_fuzz(be, monkeypatch):
    P.check_bwt_sort_fuzz(be, monkeypatch, cases=40)      # (the MI355X suite runs 120 larger ones)


def test_rank_pipe_under_ans1_decoder(be, monkeypatch):
    # (the emulator runs a small matrix: every form once on the pipeline of the bench, the bare RANK+ZRLT sequence through the fused chain and the fall-back; the MI355X suite runs all of it)
    P.check_rank_pipe(be, monkeypatch, sizes=((30000, 1 << 14), (1000, 1024)), seeds=(5,), seqs=("BWT+RANK+ZRLT",))
    P.check_rank_pipe(be, monkeypatch, sizes=((20000, 1 << 13),), seeds=(6,), seqs=("RANK+ZRLT",), forms=("pipe", "regular", "two_groups"))


def test_rank_chain_variants(be, monkeypatch):
    P.check_rank_chain_variants(be, monkeypatch, max_len=4100, bwt_len=12000)


def test_rank_inverse_patterns(be):
    P.check_rank_inverse_patterns(be, scale=1)


def test_srt_chain_form(be):
    P.check_srt_chain_form(be)


def test_utf_streams(be):
    P.check_utf_streams(be)


def test_own_sort_scan_select_kernels(be, monkeypatch):
    """The hand-written radix sort / max-scan / select of prims.hip under the emulator (the other BWT cases of this suite take the plain
    loops of prims.h to stay fast): BWT objects of every input shape, a multi-block stream, the inverse BWT's u32 sort."""
    monkeypatch.setenv("KNZ_EMU_PRIMS", "kernels")
    P.check_transform(be, "BWT", max_len=70000)
    P.check_stream(be, "BWT+RANK+ZRLT", "ANS0", 1 << 14, 100000)
    P.check_stream(be, "BWT", "NONE", 4096, 4096 * 5 + 77)


def test_ans1_encode_in_groups(be, monkeypatch):
    """The order-1 rANS encoder's workspace is sized to a group of blocks, not to the batch: 7 blocks through groups of 2."""
    monkeypatch.setenv("KNZ_ANS1_GROUP_BLOCKS", "2")
    P.check_stream(be, "NONE", "ANS1", 1 << 14, 7 * (1 << 14) - 100)
    P.check_stream(be, "BWT+RANK+ZRLT", "ANS1", 1 << 14, 5 * (1 << 14) + 33)


def test_lz_inverse_forms(be, monkeypatch):
    """Parallel LZ inverse (token scans + source map, lz_inv_par.hip) and the one-wave kernel it leaves damaged blocks to."""
    P.check_lz_inverse_forms(be, monkeypatch)


def test_text_stream_through_foreign_handle(be):
    P.check_text_foreign_handle(be)


def test_lz_forward_forms(be, monkeypatch):
    """Segment-parallel LZ parse (fixed point over segment entry states and hole maps, lz_fwd_seg.hip) against the two one-wave forms."""
    P.check_lz_forward_forms(be, monkeypatch, segs=(256, 512))


def test_lz_streams_small_segments(be, monkeypatch):
    """Whole streams with the parse cut into 512-position segments: many blocks x many segments, ragged last block."""
    monkeypatch.setenv("KNZ_LZ_SEG", "512")
    P.check_stream(be, "LZ", "ANS0", 1 << 14, 50000)
    P.check_stream(be, "LZX", "HUFFMAN", 1 << 13, 20000)
    P.check_stream(be, "LZ", "NONE", 4096, 4096 * 3 + 100)


def test_lz_first_form(be, monkeypatch):
    """KNZ_LZ_CHAIN: the parse that keeps its own hash table (lz.hip), the cross-check of the table-free forms."""
    monkeypatch.setenv("KNZ_LZ_CHAIN", "1")
    P.check_transform(be, "LZ", max_len=60000)
    P.check_transform(be, "LZX", max_len=60000)


def test_text_transform_and_streams(be):
    P.check_text(be, n=40_000, bs_stream=1 << 14,
                 streams=(("TEXT", "NONE"), ("TEXT+UTF+BWT+RANK+ZRLT", "ANS0"), ("TEXT+UTF", "HUFFMAN"), ("UTF+TEXT", "ANS1"), ("TEXT+TEXT", "FPAQ")))


def test_text_damaged_input(be, monkeypatch):
    P.check_text_damaged(be, trials=40, n=8000)
    monkeypatch.setenv("KNZ_TEXT_CHAIN", "1")
    P.check_text_damaged(be, trials=40, n=8000, seed=2)


def test_text_one_lane_scan(be, monkeypatch):
    monkeypatch.setenv("KNZ_TEXT_CHAIN", "1")
    P.check_text(be, n=12_000, chain=True, streams=())


def test_skip_blocks_option(be):
    P.check_skip_blocks(be, light=True)


@pytest.mark.timeout(900)
def test_differential_fuzz(be):
    P.check_fuzz(be, cases=120, seed=20260924, max_n=50000, heavy_max_n=5000)


@pytest.mark.timeout(600)
def test_corrupt_streams_come_back(be):
    P.check_corrupt_streams(be, trials=6)
