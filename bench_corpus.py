"""Deterministic synthetic corpora for bench.py and the large-size tests (no silesia/enwik on disk, no network).

S-silesia: 211,957,760 bytes in 12 member-shaped segments sized like silesia.tar's members:
text (word-dictionary, Zipf), executable-like (4-byte records, ~30 % zeros), 16-bit image-like (random walk),
highly repetitive records (templates + mutations), fixed-width DB rows with skewed columns.
Generator: numpy PCG64 seeded per segment; same bytes on every run/box with this image's numpy.
"""
import numpy as np

SILESIA_SIZE = 211_957_760
# (name, size, kind, seed)
_MEMBERS = [
    ("dickens", 10_192_446, "text", 1), ("mozilla", 51_220_480, "exe", 2), ("mr", 9_970_564, "img16", 3),
    ("nci", 33_553_445, "records", 4), ("ooffice", 6_152_192, "exe", 5), ("osdb", 10_085_684, "db", 6),
    ("reymont", 6_627_202, "text", 7), ("samba", 21_606_400, "source", 8), ("sao", 7_251_944, "db", 9),
    ("webster", 41_458_703, "text", 10), ("xml", 5_345_280, "records", 11), ("x-ray", 8_474_240, "img16", 12),
]


def _words(rng, count, alphabet):
    lens = rng.integers(2, 11, count)
    letters = rng.choice(alphabet, size=int(lens.sum()), p=None)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    return letters.astype(np.uint8), starts, lens


def _text(rng, n, dict_size=30000, source=False):
    alpha = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8)
    p = np.arange(1, len(alpha) + 1, dtype=np.float64) ** -0.9
    p /= p.sum()
    lens = rng.integers(2, 11, dict_size)
    letters = rng.choice(alpha, size=int(lens.sum()), p=p)
    starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
    seps = np.frombuffer(b"      \n,.;(){}=_" if source else b"        \n\n,.", dtype=np.uint8)
    out = np.empty(n + 64, dtype=np.uint8)
    pos = 0
    zipf_p = np.arange(1, dict_size + 1, dtype=np.float64) ** -1.05
    zipf_p /= zipf_p.sum()
    while pos < n:
        m = 400_000
        idx = rng.choice(dict_size, size=m, p=zipf_p)
        wl = lens[idx] + 1
        tot = int(wl.sum())
        off = np.concatenate(([0], np.cumsum(wl)[:-1]))
        within = np.arange(tot) - np.repeat(off, wl)
        src = np.repeat(starts[idx], wl) + within
        buf = np.empty(tot, dtype=np.uint8)
        is_sep = within == np.repeat(wl - 1, wl)
        buf[~is_sep] = letters[src[~is_sep]]
        buf[is_sep] = seps[rng.integers(0, len(seps), int(is_sep.sum()))]
        take = min(tot, n - pos)
        out[pos:pos + take] = buf[:take]
        pos += take
    return out[:n]


def _exe(rng, n):
    m = (n + 3) // 4
    op_p = np.arange(1, 65, dtype=np.float64) ** -1.2
    op_p /= op_p.sum()
    ops = (rng.choice(64, size=m, p=op_p) * 3 + 0x40).astype(np.uint8)
    modrm = (rng.choice(32, size=m, p=(np.arange(1, 33.0) ** -0.8) / (np.arange(1, 33.0) ** -0.8).sum()) * 8).astype(np.uint8)
    imm_lo = rng.integers(0, 256, m).astype(np.uint8)
    imm_lo[rng.random(m) < 0.35] = 0
    imm_hi = rng.integers(0, 256, m).astype(np.uint8)
    imm_hi[rng.random(m) < 0.8] = 0
    rec = np.stack([ops, modrm, imm_lo, imm_hi], axis=1).reshape(-1)
    # zero pages / padding runs
    for _ in range(max(1, n // 2_000_000)):
        s = int(rng.integers(0, max(1, n - 70000)))
        rec[s:s + int(rng.integers(4096, 65536))] = 0
    return rec[:n]


def _img16(rng, n):
    m = (n + 1) // 2
    walk = np.cumsum(rng.integers(-6, 7, m)).astype(np.int64)
    lo = (walk & 0xFF).astype(np.uint8)
    hi = ((walk >> 8) & 0x0F).astype(np.uint8)
    return np.stack([lo, hi], axis=1).reshape(-1)[:n]


def _records(rng, n, width=200, templates=64):
    t = rng.integers(32, 127, (templates, width)).astype(np.uint8)
    t[:, -1] = 10
    rows = (n + width - 1) // width
    which = rng.integers(0, templates, rows)
    out = t[which].copy()
    mut = rng.random(out.shape) < 0.05
    out[mut] = rng.integers(48, 58, int(mut.sum())).astype(np.uint8)
    return out.reshape(-1)[:n]


def _db(rng, n, width=64):
    rows = (n + width - 1) // width
    cols = []
    for c in range(width):
        k = [2, 4, 16, 256][c % 4]
        p = np.arange(1, k + 1, dtype=np.float64) ** -1.5
        p /= p.sum()
        cols.append(rng.choice(k, size=rows, p=p).astype(np.uint8))
    return np.stack(cols, axis=1).reshape(-1)[:n]


def _segment(kind, n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "text":
        return _text(rng, n)
    if kind == "source":
        return _text(rng, n, dict_size=4000, source=True)
    if kind == "exe":
        return _exe(rng, n)
    if kind == "img16":
        return _img16(rng, n)
    if kind == "records":
        return _records(rng, n)
    if kind == "db":
        return _db(rng, n)
    raise ValueError(kind)


def s_silesia(size=SILESIA_SIZE):
    """Returns a uint8 numpy array of exactly `size` bytes (members scaled down proportionally if size < full)."""
    scale = size / SILESIA_SIZE
    parts = []
    total = 0
    for name, sz, kind, seed in _MEMBERS:
        n = max(1, int(sz * scale))
        parts.append(_segment(kind, n, seed))
        total += n
    out = np.concatenate(parts)
    if len(out) < size:   # tar padding
        out = np.concatenate([out, np.zeros(size - len(out), dtype=np.uint8)])
    return np.ascontiguousarray(out[:size])


def s_enwik(size=1_000_000_000, seed=20):
    """S-enwik: XML-ish text, seed 20. Wiki-dump shaped: <page> records with a few tag lines (title, id, timestamp,
    contributor) around a body of dictionary text with [[links]], ''markup'' and &quot; entities. Generated in 16 MB pieces
    (PCG64 seeded per piece) so that any prefix is reproducible and 10^9 bytes never need more than one piece of scratch."""
    piece = 16_000_000
    parts, made, k = [], 0, 0
    while made < size:
        n = min(piece, size - made)
        rng = np.random.Generator(np.random.PCG64(seed * 1000 + k))
        body = _text(rng, n + 4096, dict_size=50000)
        out = body[:n].copy()
        # sprinkle markup: every ~2.5 KB a page header, every ~90 bytes a link / quote / entity
        hdr = np.frombuffer(b"  </revision>\n  </page>\n  <page>\n    <title>", dtype=np.uint8)
        mid = np.frombuffer(b"</title>\n    <id>", dtype=np.uint8)
        tail = np.frombuffer(b"</id>\n    <revision>\n      <timestamp>2006-03-03T", dtype=np.uint8)
        txt = np.frombuffer(b":00Z</timestamp>\n      <contributor>\n        <username>", dtype=np.uint8)
        txt2 = np.frombuffer(b"</username>\n      </contributor>\n      <text xml:space=\"preserve\">", dtype=np.uint8)
        pos = int(rng.integers(0, 1500))
        pid = 1000 + 7919 * k
        while pos + 400 < n:
            cur = pos
            for frag in (hdr, None, mid, str(pid).encode(), tail, f"{pid % 24:02d}:{pid % 60:02d}".encode(), txt, None, txt2):
                if frag is None:
                    ln = int(rng.integers(6, 24))
                    cur += ln                                   # keep the body's words as title / user name
                    continue
                f = np.frombuffer(frag, dtype=np.uint8) if isinstance(frag, bytes) else frag
                out[cur:cur + len(f)] = f
                cur += len(f)
            pid += int(rng.integers(1, 40))
            pos = cur + int(rng.integers(800, 4200))
        marks = rng.integers(40, 140, n // 90 + 1).cumsum()
        marks = marks[marks < n - 16]
        kinds = rng.integers(0, 4, len(marks))
        for m, kd in zip(marks.tolist(), kinds.tolist()):
            if kd == 0:
                out[m:m + 2] = (91, 91); ln = 5 + (m % 13); out[m + ln:m + ln + 2] = (93, 93)
            elif kd == 1:
                out[m:m + 2] = (39, 39); ln = 4 + (m % 9); out[m + ln:m + ln + 2] = (39, 39)
            elif kd == 2:
                out[m:m + 6] = np.frombuffer(b"&quot;", dtype=np.uint8)
            else:
                out[m] = 10
        parts.append(out)
        made += n
        k += 1
    return np.ascontiguousarray(np.concatenate(parts)[:size])


def s_rand(n, seed=101):
    return np.random.Generator(np.random.PCG64(seed)).integers(0, 256, n, dtype=np.uint8)


def s_zero(n):
    return np.zeros(n, dtype=np.uint8)


def s_ramp(n):
    return (np.arange(n, dtype=np.uint32) & 0xFF).astype(np.uint8)   # BWT_test.go:73-84 byte(i) ramp
