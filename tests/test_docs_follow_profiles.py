"""The round's tables quote the committed profiles: the figures of BASELINE.md / README.md for the final run must be the ones in
profiles/r06_final_config_*_bench.json (a table that drifts from its source is worse than no table)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(cfg):
    return json.loads(open(os.path.join(ROOT, "profiles", f"r06_final_config_{cfg}_bench.json")).read().strip().splitlines()[-1])


def _fmt(v):
    """the way the tables write a rate: 354.9, 1,949, 86,348"""
    return f"{v:.1f}" if v < 1000 else f"{round(v):,}"


def test_baseline_table_quotes_the_final_profiles():
    text = open(os.path.join(ROOT, "BASELINE.md")).read()
    for cfg in ("bwt", "l5", "lz", "huffman", "ans0"):
        d = _line(cfg)
        assert d["roofline"] and d["cpu_baseline"] and d["vs_baseline"] is None and d["dtype"] == "u8" and d["n_gpus"] == 1, cfg
        assert d.get("bit_exact_vs_oracle") is True and d.get("roundtrip_ok") is True, cfg
        assert d.get("bit_exact_vs_reference") is True, (cfg, "the timed stream must equal the reference-written one (fullsize_manifest.json)")
        for key in ("value",):
            assert _fmt(d[key]) in text, (cfg, key, _fmt(d[key]))
        for key in ("encode_MBps", "decode_MBps"):
            assert _fmt(d[key]) in text, (cfg, key, _fmt(d[key]))


def test_readme_quotes_the_default_configuration():
    text = open(os.path.join(ROOT, "README.md")).read()
    d = _line("bwt")
    assert _fmt(d["value"]) in text and str(round(d["encode_MBps"])) in text and str(round(d["decode_MBps"])) in text
    assert d["config"]["workload"].startswith("BASELINE.json configs[3]")
    assert d["cpu_baseline"]["kind"] == "reference" and "port" in d["cpu_baseline"]


def test_design_is_reviewable_and_quotes_the_final_profiles():
    """DESIGN.md stays a document somebody can review (round-5 verdict: 136 KB in 641 lines was not): at most 300 lines, no line beyond 400 characters outside
    tables; and its results table carries the figures of the committed lines."""
    lines = open(os.path.join(ROOT, "DESIGN.md")).read().splitlines()
    assert len(lines) <= 300, len(lines)
    assert all(len(l) <= 400 or l.startswith("|") for l in lines), max(len(l) for l in lines if not l.startswith("|"))
    text = "\n".join(lines)
    for cfg in ("bwt", "l5", "lz", "huffman", "ans0"):
        d = _line(cfg)
        assert _fmt(d["value"]) in text and _fmt(d["encode_MBps"]) in text and _fmt(d["decode_MBps"]) in text, cfg
    assert os.path.exists(os.path.join(ROOT, "docs", "HISTORY.md"))
