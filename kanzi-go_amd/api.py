"""ctypes binding of libknz_gpu.so + mirrors of the kanzi-go interfaces for the hot path."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB = None
_LIB_PATH = None

# v2/transform/Factory.go:31-53 ; v2/entropy/EntropyCodecFactory.go:26-42
_TNAMES = {"NONE": 0, "BWT": 1, "LZ": 3, "ZRLT": 6, "MTFT": 7, "RANK": 8, "SRT": 13, "LZP": 14, "LZX": 16, "UTF": 17, "TEXT": 10}
_ENAMES = {"NONE": 0, "HUFFMAN": 1, "FPAQ": 2, "ANS0": 5, "ANS1": 8}


def transform_type(name):
    """transform.GetType (v2/transform/Factory.go:289-328)."""
    if isinstance(name, int):
        return name
    res, shift = 0, 42
    for tok in name.upper().split("+"):
        t = _TNAMES[tok]
        if t:
            res |= t << shift
            shift -= 6
    return res


def entropy_type(name):
    """entropy.GetType (v2/entropy/EntropyCodecFactory.go:173-206)."""
    return name if isinstance(name, int) else _ENAMES[name.upper()]


class KnzError(RuntimeError):
    """Mirrors io.IOError{msg, code} (v2/io/CompressedStream.go:56-60); code is a kanzi.ERR_* value."""

    def __init__(self, code, msg=""):
        super().__init__(f"kanzi error {code}: {msg}")
        self.code = code


class _Cfg(C.Structure):
    _fields_ = [("transform", C.c_uint64), ("entropy", C.c_uint32), ("block_size", C.c_uint32),
                ("checksum_bits", C.c_uint32), ("bs_version", C.c_uint32), ("device", C.c_int32), ("flags", C.c_uint32)]


class _Block(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_len", C.c_uint32), ("dst", C.c_void_p), ("dst_cap", C.c_uint32),
                ("out_bits", C.c_uint64), ("post_len", C.c_uint32), ("skip_flags", C.c_uint8), ("mode", C.c_uint8),
                ("reserved", C.c_uint16), ("checksum", C.c_uint64), ("status", C.c_int32), ("reserved2", C.c_int32)]


def library_path():
    return os.environ.get("KNZ_GPU_LIB", os.path.join(_HERE, "libknz_gpu.so"))


def build_library(verbose=False):
    """hipcc cross-compiles the gfx950 library in-tree (works without a GPU)."""
    src = os.path.join(_HERE, "csrc", "knz_gpu.hip")
    out = os.path.join(_HERE, "libknz_gpu.so")
    deps = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    deps.append(os.path.join(_ROOT, "include", "knz_gpu.h"))
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-variable", "-Wno-unused-value",
           "-o", out, src]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


def load_library(path=None):
    """Loads libknz_gpu.so. Raises if it is missing: the product path never falls back to the CPU."""
    global _LIB, _LIB_PATH
    path = path or library_path()
    if _LIB is not None and _LIB_PATH == path:
        return _LIB
    if not os.path.exists(path):
        raise KnzError(4, f"{path} not found: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                          "there is no CPU fallback")
    L = C.CDLL(path)
    vp, u8p, u64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64)
    L.knz_open.argtypes = [C.POINTER(_Cfg), C.POINTER(vp)]
    L.knz_open_devices.argtypes = [C.POINTER(_Cfg), C.POINTER(C.c_int32), C.c_int, C.POINTER(vp)]
    L.knz_device_count.argtypes = []
    L.knz_lane_count.argtypes = [vp]
    L.knz_last_lane_times.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int]
    L.knz_close.argtypes = [vp]
    L.knz_last_error.argtypes = [vp]
    L.knz_last_error.restype = C.c_char_p
    L.knz_encode_blocks.argtypes = [vp, C.POINTER(_Block), C.c_int]
    L.knz_decode_blocks.argtypes = [vp, C.POINTER(_Block), C.c_int]
    L.knz_dev_compress.argtypes = [vp, vp, C.c_uint64, C.c_int64, vp, C.c_uint64, u64p, vp]
    L.knz_dev_decompress.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, u64p, vp]
    L.knz_dev_compress_blocks.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, u64p, vp]
    L.knz_dev_decompress_blocks.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, u64p, vp]
    L.knz_dev_assemble.argtypes = [vp, C.c_int64, C.POINTER(vp), u64p, C.c_int, vp, C.c_uint64, u64p, vp]
    L.knz_transform_forward.argtypes = [vp, C.c_uint64, u8p, C.c_uint32, u8p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.knz_transform_inverse.argtypes = [vp, C.c_uint64, u8p, C.c_uint32, u8p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.knz_max_encoded_len.argtypes = [C.c_uint64, C.c_uint32]
    L.knz_max_encoded_len.restype = C.c_uint32
    L.knz_entropy_encode.argtypes = [vp, C.c_uint32, u8p, C.c_uint32, u8p, C.c_uint64, u64p]
    L.knz_entropy_decode.argtypes = [vp, C.c_uint32, u8p, C.c_uint64, u8p, C.c_uint32, u64p]
    L.knz_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.knz_last_counter.argtypes = [vp, C.c_int, C.POINTER(C.c_uint64)]
    L.knz_last_counter.restype = C.c_int
    L.knz_supports.argtypes = [C.c_uint64, C.c_uint32]
    L.knz_last_kernel_times.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_float), C.c_int]
    _LIB, _LIB_PATH = L, path
    return L


def _u8(a):
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


class Codec:
    """One GPU batch scheduler handle (knz_open): what an io.Writer/io.Reader owns in the drop-in."""

    def __init__(self, transform="NONE", entropy="NONE", block_size=4 << 20, checksum_bits=0, device=-1, lib=None, skip_blocks=False, devices=None):
        """skip_blocks = the CLI's -s / ctx["skipBlocks"] (KNZ_FLAG_SKIP_BLOCKS): incompressible blocks become copy blocks.
        devices = a list of HIP ordinals (knz_open_devices): one lane each, the block batches fan out over them; an ordinal may repeat."""
        self.L = load_library(lib)
        self.cfg = _Cfg(transform_type(transform), entropy_type(entropy), block_size, checksum_bits, 6, device, 1 if skip_blocks else 0)
        self.h = C.c_void_p()
        if devices is not None:
            ords = (C.c_int32 * len(devices))(*devices)
            rc = self.L.knz_open_devices(C.byref(self.cfg), ords, len(devices), C.byref(self.h))
        else:
            rc = self.L.knz_open(C.byref(self.cfg), C.byref(self.h))
        if rc:
            raise KnzError(rc, "knz_open failed: " + self.L.knz_last_error(None).decode() + " (no CPU fallback exists)")

    def lane_times(self):
        """[(device ordinal, blocks taken, wall-clock ms)] per lane of the last batch call of a knz_open_devices handle."""
        n = self.L.knz_lane_count(self.h)
        dev, blk, ms = (C.c_int32 * n)(), (C.c_int32 * n)(), (C.c_float * n)()
        k = self.L.knz_last_lane_times(self.h, dev, blk, ms, n)
        return [(int(dev[i]), int(blk[i]), float(ms[i])) for i in range(k)]

    def close(self):
        if self.h:
            self.L.knz_close(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc:
            raise KnzError(rc, self.L.knz_last_error(self.h).decode())

    # ---- device-resident whole-stream calls (pointers are integers: tensor.data_ptr()) ----
    def dev_compress(self, d_src, n, d_dst, dst_cap, header_input_size=None, stream=0):
        out = C.c_uint64()
        hs = n if header_input_size is None else header_input_size
        self._chk(self.L.knz_dev_compress(self.h, d_src, n, hs, d_dst, dst_cap, C.byref(out), stream))
        return out.value

    def dev_decompress(self, d_src, n_bytes, d_dst, dst_cap, stream=0):
        out = C.c_uint64()
        self._chk(self.L.knz_dev_decompress(self.h, d_src, n_bytes, d_dst, dst_cap, C.byref(out), stream))
        return out.value

    def dev_compress_blocks(self, d_src, n, d_dst, dst_cap, stream=0):
        out = C.c_uint64()
        self._chk(self.L.knz_dev_compress_blocks(self.h, d_src, n, d_dst, dst_cap, C.byref(out), stream))
        return out.value

    def dev_decompress_blocks(self, d_src, n_bits, d_dst, dst_cap, stream=0):
        out = C.c_uint64()
        self._chk(self.L.knz_dev_decompress_blocks(self.h, d_src, n_bits, d_dst, dst_cap, C.byref(out), stream))
        return out.value

    def dev_assemble(self, header_input_size, segments, segment_bits, d_dst, dst_cap, stream=0):
        n = len(segments)
        segs = (C.c_void_p * n)(*segments)
        bits = (C.c_uint64 * n)(*segment_bits)
        out = C.c_uint64()
        self._chk(self.L.knz_dev_assemble(self.h, header_input_size, segs, bits, n, d_dst, dst_cap, C.byref(out), stream))
        return out.value

    def last_counter(self, cid=0):
        v = C.c_uint64()
        self._chk(self.L.knz_last_counter(self.h, cid, C.byref(v)))
        return v.value

    def last_timing(self):
        t = (C.c_float * 4)()
        n = self.L.knz_last_timing(self.h, t, 4)
        return [t[i] for i in range(n)]

    def last_kernel_times(self):
        """[(kernel name, ms)] of the probed launches of the last device batch (HIP events on the launch stream)."""
        names = C.create_string_buffer(16384)
        ms = (C.c_float * 128)()
        n = self.L.knz_last_kernel_times(self.h, names, 16384, ms, 128)
        nm = names.value.decode().split("\n")
        return [(nm[i], float(ms[i])) for i in range(n)]


class BlockBatch:
    """Writer.processBlock / Reader.processBlock re-pointed at the GPU (host buffers in, host buffers out)."""

    def __init__(self, codec: Codec):
        self.c = codec

    def encode(self, blocks):
        """blocks: list of bytes. Returns [(block_local_stream_bytes, written_bits, mode, post_len, skip_flags)] like
        encodingTask.encode up to obs.Close() (v2/io/CompressedStream.go:729-914)."""
        n = len(blocks)
        arr = (_Block * n)()
        keep = []
        for i, b in enumerate(blocks):
            a, _ = _u8(b)
            cap = int(self.c.L.knz_max_encoded_len(self.c.cfg.transform, len(a))) * 2 + 262144
            o = np.zeros(cap, dtype=np.uint8)
            keep.append((a, o))
            arr[i].src = a.ctypes.data
            arr[i].src_len = len(a)
            arr[i].dst = o.ctypes.data
            arr[i].dst_cap = cap
        self.c._chk(self.c.L.knz_encode_blocks(self.c.h, arr, n))
        return [(keep[i][1][: (arr[i].out_bits + 7) // 8].tobytes(), int(arr[i].out_bits), int(arr[i].mode), int(arr[i].post_len),
                 int(arr[i].skip_flags)) for i in range(n)]

    def decode(self, payloads):
        """payloads: list of block-local streams. Returns the decoded blocks (decodingTask.decode :1875-2011)."""
        n = len(payloads)
        arr = (_Block * n)()
        keep = []
        cap = self.c.cfg.block_size + max(512, self.c.cfg.block_size >> 4)
        for i, b in enumerate(payloads):
            a, _ = _u8(b)
            o = np.zeros(cap, dtype=np.uint8)
            keep.append((a, o))
            arr[i].src = a.ctypes.data
            arr[i].src_len = len(a)
            arr[i].dst = o.ctypes.data
            arr[i].dst_cap = cap
        self.c._chk(self.c.L.knz_decode_blocks(self.c.h, arr, n))
        return [keep[i][1][: arr[i].out_bits].tobytes() for i in range(n)]


class EntropyEncoder:
    """kanzi.EntropyEncoder (v2/Definitions.go:154-165) over knz_entropy_encode: Write returns the bit string the
    Go shim hands to obs.WriteArray."""

    def __init__(self, codec: Codec, etype):
        self.c, self.t = codec, entropy_type(etype)

    def write(self, block):
        a, p = _u8(block)
        cap = len(a) * 2 + 262144
        out = np.zeros(cap, dtype=np.uint8)
        bits = C.c_uint64()
        self.c._chk(self.c.L.knz_entropy_encode(self.c.h, self.t, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(bits)))
        return out[: (bits.value + 7) // 8].tobytes(), bits.value


class EntropyDecoder:
    """kanzi.EntropyDecoder (v2/Definitions.go:168-179) over knz_entropy_decode."""

    def __init__(self, codec: Codec, etype):
        self.c, self.t = codec, entropy_type(etype)

    def read(self, payload, n):
        a, p = _u8(payload)
        out = np.zeros(max(n, 1), dtype=np.uint8)
        used = C.c_uint64()
        self.c._chk(self.c.L.knz_entropy_decode(self.c.h, self.t, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), n, C.byref(used)))
        return out[:n].tobytes(), used.value


class ByteTransform:
    """kanzi.ByteTransform (v2/Definitions.go:78-91) over knz_transform_forward/inverse. forward() returns None
    when the transform declines (a Forward error means "skip", v2/transform/Sequence.go:86-91)."""

    def __init__(self, codec: Codec, ttype):
        self.c = codec
        self.t = _TNAMES[ttype.upper()] if isinstance(ttype, str) else ttype

    def max_encoded_len(self, n):
        return int(self.c.L.knz_max_encoded_len(self.t << 42, n))

    def forward(self, src):
        a, p = _u8(src)
        cap = self.max_encoded_len(len(a)) + 64
        out = np.zeros(cap, dtype=np.uint8)
        n = C.c_uint32()
        rc = self.c.L.knz_transform_forward(self.c.h, self.t, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n))
        if rc == -1:
            return None
        self.c._chk(rc)
        return out[: n.value].tobytes()

    def inverse(self, src, cap):
        a, p = _u8(src)
        out = np.zeros(max(cap, 1), dtype=np.uint8)
        n = C.c_uint32()
        self.c._chk(self.c.L.knz_transform_inverse(self.c.h, self.t, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
        return out[: n.value].tobytes()
