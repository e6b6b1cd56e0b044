#!/usr/bin/env python3
"""Stress of the segment-parallel LZ / LZX parse (lz_fwd_seg.hip, round 6: one lane per segment, re-runs chosen from the data, border guards): random mixes of
the shapes that make holes and cross-segment dependencies (executable-like, records, text, incompressible stretches, zero runs, periodic data) at
segment sizes from 64 positions up, both forms (lanes / KNZ_LZS_WAVES), every result == the oracle's LZCodec.Forward.
Arguments: seconds to spend (default 120), first seed (default 1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench_corpus as bc
import parity_cases as P
import oracle_lib as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
P.K.build_library()
be = P.GpuBackend()
c = P.K.Codec("NONE", "NONE", 4 << 20, lib=be.lib)
t0 = time.time(); cases = 0; fallbacks = 0; max_rounds = 0
while time.time() - t0 < budget:
    r = np.random.default_rng(seed)
    parts = []
    for _ in range(int(r.integers(2, 9))):
        kind = int(r.integers(0, 7)); n = int(r.integers(200, 200000))
        if kind == 0: parts.append(bc._segment("exe", n, int(r.integers(1, 1000))).tobytes())
        elif kind == 1: parts.append(bc._segment("records", n, int(r.integers(1, 1000))).tobytes())
        elif kind == 2: parts.append(bc._segment("text", n, int(r.integers(1, 1000))).tobytes())
        elif kind == 3: parts.append(r.integers(0, 256, n, dtype=np.uint8).tobytes())
        elif kind == 4: parts.append(bytes(n // 4))
        elif kind == 5: per = r.integers(0, 256, int(r.integers(3, 300)), dtype=np.uint8).tobytes(); parts.append((per * (n // len(per) + 1))[:n])
        else: parts.append(parts[int(r.integers(0, len(parts)))][: n] if parts else bytes(n))
    data = b"".join(parts)[: 3_500_000]
    for tname in ("LZ", "LZX"):
        t = P.K.ByteTransform(c, tname)
        want = O.transform_forward(P._TID[tname], data)
        for seg in (64, 128, 256, 768, 2048):
            for waves in (False, True, None):                      # None: lanes with a short list of moved map words ("too many moved" in most rounds)
                if waves is None and seg not in (128, 768): continue
                os.environ["KNZ_LZ_SEG"] = str(seg)
                if waves: os.environ["KNZ_LZS_WAVES"] = "1"
                else: os.environ.pop("KNZ_LZS_WAVES", None)
                if waves is None: os.environ["KNZ_LZS_CHG_CAP"] = "1024"
                else: os.environ.pop("KNZ_LZS_CHG_CAP", None)
                got = t.forward(data)
                assert got == want, (seed, tname, seg, waves, len(data))
                cases += 1
                fallbacks += c.last_counter(4); max_rounds = max(max_rounds, c.last_counter(5))
    print(f"seed {seed}: ok ({cases} parses, {fallbacks} blocks left to the one-wave kernel, most rounds {max_rounds}, {time.time() - t0:.0f} s)", flush=True)
    seed += 1
os.environ.pop("KNZ_LZ_SEG", None); os.environ.pop("KNZ_LZS_WAVES", None); os.environ.pop("KNZ_LZS_CHG_CAP", None)
