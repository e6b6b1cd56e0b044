"""ctypes binding of oracle/_ref/libknz_ref.so — TEST INFRASTRUCTURE ONLY.

oracle/_ref is the reference itself: kanzi-go's own source files, translated mechanically to C++ by tools/go2cpp and compiled by
`make -C oracle _ref` (needs /root/reference; the generated source and the library are git-ignored, the built library travels to
the GPU box with the repository snapshot). Same call shapes as oracle_lib.py so that the tests can cross the two checkers.
Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libknz_ref.so")
REFERENCE = "/root/reference/v2"
_LIB = None


def can_build():
    return os.path.isdir(REFERENCE)


def available():
    return os.path.exists(REF_SO) or can_build()


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "_ref"])


def lib():
    global _LIB
    if _LIB is None:
        if can_build():
            build()                                  # (make: a no-op when the library is newer than its sources)
        if not os.path.exists(REF_SO):
            raise RuntimeError("oracle/_ref/libknz_ref.so is missing and /root/reference is not here to build it from")
        L = C.CDLL(REF_SO)
        u8p = C.POINTER(C.c_uint8)
        u64p = C.POINTER(C.c_uint64)
        L.kref_last_error.restype = C.c_char_p
        L.kref_entropy_encode.argtypes = [C.c_uint32, u8p, C.c_uint64, u8p, C.c_uint64, u64p]
        L.kref_entropy_decode.argtypes = [C.c_uint32, u8p, C.c_uint64, u8p, C.c_uint64, u64p]
        L.kref_transform_forward.argtypes = [C.c_uint64, u8p, C.c_uint64, u8p, C.c_uint64, u64p]
        L.kref_transform_inverse.argtypes = [C.c_uint64, u8p, C.c_uint64, u8p, C.c_uint64, u64p]
        L.kref_sequence_forward.argtypes = [C.c_uint64, u8p, C.c_uint64, u8p, C.c_uint64, u64p, u8p]
        L.kref_sequence_inverse.argtypes = [C.c_uint64, C.c_uint8, u8p, C.c_uint64, u8p, C.c_uint64, u64p]
        L.kref_max_encoded_len.argtypes = [C.c_uint64, C.c_uint64]
        L.kref_max_encoded_len.restype = C.c_uint64
        L.kref_set_ctx.argtypes = [C.c_uint32, C.c_uint32]
        L.kref_set_data_type.argtypes = [C.c_int]
        L.kref_bwt_forward.argtypes = [u8p, C.c_uint64, u8p, u64p]
        L.kref_bwt_inverse.argtypes = [u8p, C.c_uint64, u8p, u64p]
        L.kref_xxhash32.argtypes = [u8p, C.c_uint64, C.c_uint32]
        L.kref_xxhash32.restype = C.c_uint32
        L.kref_xxhash64.argtypes = [u8p, C.c_uint64, C.c_uint64]
        L.kref_xxhash64.restype = C.c_uint64
        L.kref_magic_type.argtypes = [u8p, C.c_uint64]
        L.kref_magic_type.restype = C.c_uint32
        L.kref_entropy1024.argtypes = [u8p, C.c_uint64]
        L.kref_transform_type.argtypes = [C.c_char_p]
        L.kref_transform_type.restype = C.c_uint64
        L.kref_entropy_type.argtypes = [C.c_char_p]
        L.kref_entropy_type.restype = C.c_uint32
        _LIB = L
    return _LIB


class RefError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"_ref rc {code}: {msg}")
        self.code = code


def _u8(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    if len(a) == 0:
        a = np.zeros(1, dtype=np.uint8)[:0]
    keep = a if len(a) else np.zeros(1, dtype=np.uint8)
    return a, keep.ctypes.data_as(C.POINTER(C.c_uint8)), keep


def _chk(rc):
    if rc != 0:
        raise RefError(rc, lib().kref_last_error().decode(errors="replace"))


def entropy_encode(etype, data):
    """-> (bytes, bit_count)"""
    a, p, _k = _u8(data)
    cap = 2 * len(a) + 131072
    out = np.zeros(cap, dtype=np.uint8)
    bits = C.c_uint64()
    _chk(lib().kref_entropy_encode(etype, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(bits)))
    return out[: (bits.value + 7) // 8].tobytes(), bits.value


def entropy_decode(etype, payload, n):
    a, p, _k = _u8(payload)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    used = C.c_uint64()
    _chk(lib().kref_entropy_decode(etype, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), n, C.byref(used)))
    return out[:n].tobytes(), used.value


def set_ctx(block_size=0, entropy=None, data_type=-1):
    lib().kref_set_ctx(block_size, 0xFFFFFFFF if entropy is None else entropy)
    lib().kref_set_data_type(data_type)


def data_type():
    return lib().kref_get_data_type()


def transform_forward(t, data, cap=None):
    """one transform object; None when Forward returns an error (= the sequence skips it)"""
    a, p, _k = _u8(data)
    cap = cap if cap is not None else 2 * len(a) + 65536
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64()
    rc = lib().kref_transform_forward(t, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n))
    if rc == -1:
        return None
    _chk(rc)
    return out[: n.value].tobytes()


def transform_inverse(t, data, cap):
    a, p, _k = _u8(data)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_uint64()
    _chk(lib().kref_transform_inverse(t, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[: n.value].tobytes()


def sequence_forward(ttype, data):
    """-> (bytes, skip_flags)"""
    a, p, _k = _u8(data)
    cap = int(lib().kref_max_encoded_len(ttype, len(a))) + 65536
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64()
    sf = C.c_uint8()
    _chk(lib().kref_sequence_forward(ttype, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n), C.byref(sf)))
    return out[: n.value].tobytes(), sf.value


def sequence_inverse(ttype, skip_flags, data, cap):
    a, p, _k = _u8(data)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_uint64()
    _chk(lib().kref_sequence_inverse(ttype, skip_flags, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[: n.value].tobytes()


def bwt_forward(data):
    a, p, _k = _u8(data)
    out = np.zeros(max(len(a), 1), dtype=np.uint8)
    prim = (C.c_uint64 * 8)()
    _chk(lib().kref_bwt_forward(p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), prim))
    return out[: len(a)].tobytes(), list(prim)


def bwt_inverse(data, primary):
    a, p, _k = _u8(data)
    out = np.zeros(max(len(a), 1), dtype=np.uint8)
    prim = (C.c_uint64 * 8)(*primary)
    _chk(lib().kref_bwt_inverse(p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), prim))
    return out[: len(a)].tobytes()


def xxhash32(data, seed):
    a, p, _k = _u8(data)
    return int(lib().kref_xxhash32(p, len(a), seed))


def xxhash64(data, seed):
    a, p, _k = _u8(data)
    return int(lib().kref_xxhash64(p, len(a), seed))


def magic_type(data):
    a, p, _k = _u8(data)
    return int(lib().kref_magic_type(p, len(a)))


def entropy1024(data):
    a, p, _k = _u8(data)
    return int(lib().kref_entropy1024(p, len(a)))


def transform_type(name):
    return int(lib().kref_transform_type(name.encode()))


def entropy_type(name):
    return int(lib().kref_entropy_type(name.encode()))


def compress(data, transform="NONE", entropy="NONE", block_size=4 << 20, checksum_bits=0, jobs=1, header_size=None, skip_blocks=False):
    """the reference's own Writer (io/CompressedStream.go) over `data` -> the .knz stream; header_size: ctx["fileSize"] (default len(data), -1 = absent)"""
    L = lib()
    L.kref_compress.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, C.c_int,
                                C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint64)]
    a, p, _k = _u8(data)
    cap = len(a) + len(a) // 2 + (1 << 20)
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64()
    fs = len(a) if header_size is None else header_size
    _chk(L.kref_compress(p, len(a), transform.encode(), entropy.encode(), block_size, checksum_bits, jobs, fs, 1 if skip_blocks else 0,
                         out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[: n.value].tobytes()


def decompress(stream, cap, jobs=1):
    L = lib()
    L.kref_decompress.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint64)]
    a, p, _k = _u8(stream)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_uint64()
    _chk(L.kref_decompress(p, len(a), jobs, out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[: n.value].tobytes()


def record_events(verbosity, L=None):
    """from now on the stream calls attach a kanzi.Listener that writes every event down (ctx["verbosity"] = verbosity; < 0: stop)"""
    (L or lib()).kref_record_events(verbosity)


def event_log(L=None):
    """the events of the last stream call, one per line: "type id size hash hashType" or "type id msg" (times left out)"""
    L = L or lib()
    L.kref_event_log.argtypes = [C.c_char_p, C.c_uint64]
    L.kref_event_log.restype = C.c_uint64
    n = L.kref_event_log(None, 0)
    buf = C.create_string_buffer(int(n) + 1)
    L.kref_event_log(buf, n + 1)
    return buf.value.decode().splitlines()
