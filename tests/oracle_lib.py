"""ctypes binding of oracle/libknz_oracle.so — TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of the kanzi-go hot path (see oracle/*.hpp for the
reference file:line each function follows). Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None

# transform / entropy ids (v2/transform/Factory.go:31-53, v2/entropy/EntropyCodecFactory.go:26-42)
T_NONE, T_BWT, T_LZ, T_ZRLT, T_MTFT, T_RANK, T_LZX = 0, 1, 3, 6, 7, 8, 16
T_SRT, T_LZP, T_UTF, T_TEXT = 13, 14, 17, 10
E_NONE, E_HUFFMAN, E_FPAQ, E_ANS0, E_ANS1 = 0, 1, 2, 5, 8
_TNAMES = {"NONE": 0, "BWT": 1, "LZ": 3, "ZRLT": 6, "MTFT": 7, "RANK": 8, "SRT": 13, "LZP": 14, "LZX": 16, "UTF": 17, "TEXT": 10}
_ENAMES = {"NONE": 0, "HUFFMAN": 1, "FPAQ": 2, "ANS0": 5, "ANS1": 8}


def transform_type(name: str) -> int:
    """v2/transform/Factory.go:289-328 GetType: 6-bit ids packed from bit 42 downwards."""
    res, shift = 0, 42
    for tok in name.upper().split("+"):
        t = _TNAMES[tok]
        if t != 0:
            res |= t << shift
            shift -= 6
    return res


def entropy_type(name: str) -> int:
    return _ENAMES[name.upper()]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libknz_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORACLE_DIR, "libknz_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        u8p = C.POINTER(C.c_uint8)
        L.knzo_last_error.restype = C.c_char_p
        L.knzo_entropy_encode.argtypes = [C.c_uint32, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_entropy_decode.argtypes = [C.c_uint32, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_transform_forward.argtypes = [C.c_uint64, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_transform_inverse.argtypes = [C.c_uint64, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_set_ctx.argtypes = [C.c_uint32, C.c_uint32]
        L.knzo_max_encoded_len.argtypes = [C.c_uint64, C.c_uint64]
        L.knzo_max_encoded_len.restype = C.c_uint64
        L.knzo_sequence_forward.argtypes = [C.c_uint64, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64), u8p]
        L.knzo_sequence_inverse.argtypes = [C.c_uint64, C.c_uint8, u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_encode_block.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, u8p, C.c_uint64,
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), u8p, u8p, C.POINTER(C.c_uint64)]
        L.knzo_decode_block.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_uint64, u8p, C.c_uint64,
                                        C.POINTER(C.c_uint64)]
        L.knzo_compress.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_int64,
                                    u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_compress2.argtypes = [u8p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_int64, C.c_int,
                                     u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_magic_type.argtypes = [u8p, C.c_uint64]
        L.knzo_magic_type.restype = C.c_uint32
        L.knzo_entropy1024.argtypes = [u8p, C.c_uint64]
        L.knzo_log2_scaled_1024.argtypes = [C.c_uint32]
        L.knzo_log2_scaled_1024.restype = C.c_uint32
        L.knzo_decompress.argtypes = [u8p, C.c_uint64, C.c_int, u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_varint.argtypes = [C.c_uint32, u8p]
        L.knzo_varint_read.argtypes = [u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_varint_read.restype = C.c_uint32
        L.knzo_expgolomb_word.argtypes = [C.c_uint8]
        L.knzo_expgolomb_word.restype = C.c_uint32
        L.knzo_xxhash32.argtypes = [u8p, C.c_uint64, C.c_uint32]
        L.knzo_xxhash32.restype = C.c_uint32
        L.knzo_xxhash64.argtypes = [u8p, C.c_uint64, C.c_uint64]
        L.knzo_xxhash64.restype = C.c_uint64
        L.knzo_header.argtypes = [C.c_int, C.c_uint32, C.c_uint64, C.c_int64, C.c_int64, u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_normalize_frequencies.argtypes = [C.POINTER(C.c_int64), C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_int)]
        L.knzo_huffman_codes.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_uint16), u8p, C.c_uint64, C.POINTER(C.c_uint64)]
        L.knzo_bwt_forward.argtypes = [u8p, C.c_uint64, u8p, C.POINTER(C.c_uint64)]
        L.knzo_bwt_inverse.argtypes = [u8p, C.c_uint64, u8p, C.POINTER(C.c_uint64)]
        L.knzo_suffix_array.argtypes = [u8p, C.c_uint64, C.POINTER(C.c_int32)]
        L.knzo_suffix_array_divsufsort.argtypes = [u8p, C.c_uint64, C.POINTER(C.c_int32)]
        L.knzo_set_bwt_algo.argtypes = [C.c_int]
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"oracle error {code}: {msg}")
        self.code = code


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray)) else a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def _chk(rc):
    if rc != 0:
        raise OracleError(rc, lib().knzo_last_error().decode())


def entropy_encode(etype, data):
    """-> (bytes, bit_count)"""
    a, p = _u8(data)
    cap = 2 * len(a) + 131072
    out = np.zeros(cap, dtype=np.uint8)
    bits = C.c_uint64()
    _chk(lib().knzo_entropy_encode(etype, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(bits)))
    return out[: (bits.value + 7) // 8].tobytes(), bits.value


def entropy_decode(etype, payload, n):
    a, p = _u8(payload)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    used = C.c_uint64()
    _chk(lib().knzo_entropy_decode(etype, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), n, C.byref(used)))
    return out[:n].tobytes(), used.value


def set_ctx(block_size=0, entropy=None):
    """ctx["blockSize"] / ctx["entropy"] (type id) seen by single transform / block calls on this thread (TEXT reads them)."""
    lib().knzo_set_ctx(int(block_size), 0xFFFFFFFF if entropy is None else int(entropy))


def data_type():
    """ctx["dataType"] as the last transform call on this thread left it (internal/Global.go:26-40 numbering)."""
    return int(lib().knzo_get_data_type())


def transform_forward(t, data):
    """-> bytes or None when the transform declines (Forward error => skipped)."""
    a, p = _u8(data)
    cap = max(len(a) + len(a) // 8 + 64, int(lib().knzo_max_encoded_len(t << 42, len(a))) + 64)
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64()
    rc = lib().knzo_transform_forward(t, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n))
    if rc == -1:
        return None
    _chk(rc)
    return out[: n.value].tobytes()


def transform_inverse(t, data, cap):
    a, p = _u8(data)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_uint64()
    _chk(lib().knzo_transform_inverse(t, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[: n.value].tobytes()


def sequence_forward(ttype, data):
    a, p = _u8(data)
    cap = int(lib().knzo_max_encoded_len(ttype, len(a))) + 64
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64()
    sf = C.c_uint8()
    _chk(lib().knzo_sequence_forward(ttype, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n), C.byref(sf)))
    return out[: n.value].tobytes(), sf.value


def sequence_inverse(ttype, skip_flags, data, cap):
    a, p = _u8(data)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_uint64()
    _chk(lib().knzo_sequence_inverse(ttype, skip_flags, p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[: n.value].tobytes()


def encode_block(data, ttype, etype, checksum_bits=0):
    """-> dict(bits=bytes, written=int, post_len, skip_flags, mode, checksum)"""
    a, p = _u8(data)
    cap = 2 * len(a) + 65536
    out = np.zeros(cap, dtype=np.uint8)
    bits, post, sf, mode, ck = C.c_uint64(), C.c_uint32(), C.c_uint8(), C.c_uint8(), C.c_uint64()
    _chk(lib().knzo_encode_block(p, len(a), ttype, etype, checksum_bits, out.ctypes.data_as(C.POINTER(C.c_uint8)), cap,
                                 C.byref(bits), C.byref(post), C.byref(sf), C.byref(mode), C.byref(ck)))
    return dict(bits=out[: (bits.value + 7) // 8].tobytes(), written=bits.value, post_len=post.value,
                skip_flags=sf.value, mode=mode.value, checksum=ck.value)


def decode_block(payload, ttype, etype, block_size, checksum_bits=0):
    a, p = _u8(payload)
    cap = block_size + max(512, block_size >> 4)
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64()
    _chk(lib().knzo_decode_block(p, len(a), ttype, etype, checksum_bits, cap, out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[: n.value].tobytes()


def compress(data, transform="NONE", entropy="NONE", block_size=4 << 20, checksum_bits=0, jobs=1, header_size=None, skip_blocks=False):
    a, p = _u8(data)
    cap = len(a) + len(a) // 4 + 65536
    if block_size < (1 << 18):                      # order-1 headers on small blocks can outweigh the data
        cap = 2 * len(a) + 262144 * (len(a) // block_size + 2)
    out = np.zeros(cap, dtype=np.uint8)
    n = C.c_uint64()
    hs = len(a) if header_size is None else header_size
    tt = transform_type(transform) if isinstance(transform, str) else transform
    et = entropy_type(entropy) if isinstance(entropy, str) else entropy
    _chk(lib().knzo_compress2(p, len(a), tt, et, block_size, checksum_bits, jobs, hs, 1 if skip_blocks else 0,
                              out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[: n.value].tobytes()


def decompress(stream, cap, jobs=1):
    a, p = _u8(stream)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    n = C.c_uint64()
    _chk(lib().knzo_decompress(p, len(a), jobs, out.ctypes.data_as(C.POINTER(C.c_uint8)), cap, C.byref(n)))
    return out[: n.value].tobytes()


def bwt_forward(data):
    a, p = _u8(data)
    out = np.zeros(max(len(a), 1), dtype=np.uint8)
    prim = (C.c_uint64 * 8)()
    _chk(lib().knzo_bwt_forward(p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), prim))
    return out[: len(a)].tobytes(), list(prim)


def bwt_inverse(data, primary):
    a, p = _u8(data)
    out = np.zeros(max(len(a), 1), dtype=np.uint8)
    prim = (C.c_uint64 * 8)(*primary)
    _chk(lib().knzo_bwt_inverse(p, len(a), out.ctypes.data_as(C.POINTER(C.c_uint8)), prim))
    return out[: len(a)].tobytes()


def suffix_array(data):
    a, p = _u8(data)
    sa = np.zeros(max(len(a), 1), dtype=np.int32)
    _chk(lib().knzo_suffix_array(p, len(a), sa.ctypes.data_as(C.POINTER(C.c_int32))))
    return sa[: len(a)]


def suffix_array_divsufsort(data):
    """oracle/divsufsort.hpp (DivSufSort.go restated)"""
    a, p = _u8(data)
    sa = np.zeros(max(len(a), 1), dtype=np.int32)
    _chk(lib().knzo_suffix_array_divsufsort(p, len(a), sa.ctypes.data_as(C.POINTER(C.c_int32))))
    return sa[: len(a)]


def set_bwt_algo(algo):
    """1 = DivSufSort restatement (default: the reference's algorithm), 0 = SA-IS (independent cross-check)"""
    lib().knzo_set_bwt_algo(int(algo))


def bwt_kind():
    return "DivSufSort.go restated (oracle/divsufsort.hpp)" if lib().knzo_get_bwt_algo() == 1 else "SA-IS (not the reference's divsufsort)"


def huffman_codes(freqs):
    f = (C.c_int64 * 256)(*[int(x) for x in freqs])
    codes = (C.c_uint16 * 256)()
    hdr = np.zeros(4096, dtype=np.uint8)
    bits = C.c_uint64()
    _chk(lib().knzo_huffman_codes(f, codes, hdr.ctypes.data_as(C.POINTER(C.c_uint8)), 4096, C.byref(bits)))
    return np.array(codes, dtype=np.uint16), hdr[: (bits.value + 7) // 8].tobytes(), bits.value
