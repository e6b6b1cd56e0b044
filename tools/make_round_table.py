#!/usr/bin/env python3
"""The results table of a round, straight from the committed bench lines: tools/make_round_table.py r06_final -> markdown rows for BASELINE.md / DESIGN.md.
(tests/test_docs_follow_profiles.py checks that the documents quote these files.)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fmt(v):
    return f"{v:.1f}" if v < 1000 else f"{round(v):,}"


def line(tag, cfg):
    return json.loads(open(os.path.join(ROOT, "profiles", f"{tag}_config_{cfg}_bench.json")).read().strip().splitlines()[-1])


def main(tag):
    names = {"bwt": "3 (bench default): `-t BWT+RANK+ZRLT -e ANS1 -b 8m`", "l5": "`-l 5` = `-t TEXT+UTF+BWT+RANK+ZRLT -e ANS0 -b 4m`", "lz": "2: `-t LZ -e ANS0 -b 4m`",
             "huffman": "1: `-t NONE -e HUFFMAN -b 4m`", "ans0": "2, entropy half: `-t NONE -e ANS0 -b 4m`"}
    for cfg in ("bwt", "l5", "lz", "huffman", "ans0"):
        d = line(tag, cfg)
        r, c = d["roofline"], d["cpu_baseline"]
        tr = r.get("traffic_over_algorithmic")
        print(f"| {names[cfg]} | {d['config']['blocks']} | {fmt(d['encode_MBps'])} | {fmt(d['decode_MBps'])} | {fmt(d['value'])} | "
              f"{'yes' if d.get('bit_exact_vs_oracle') else 'NO'} / {'yes' if d.get('bit_exact_vs_reference') else ('-' if d.get('bit_exact_vs_reference') is None else 'NO')} | "
              f"`{r['kernel']}` {r['frac']:.2e} ({r['avg_launch_ms']:.1f} ms x {r['launches_per_step']:g}; counters {tr if tr is not None else '-'}x) | "
              f"{fmt(c['encode_MBps'])} / {fmt(c['decode_MBps'])} ({c['cores']}): {c['gpu_encode_over_cpu_encode']}x / {c['gpu_decode_over_cpu_decode']}x | "
              f"{fmt(d['host_hook_MBps']['encode'])} / {fmt(d['host_hook_MBps']['decode'])} / {fmt(d['host_hook_MBps']['round_trip'])} |")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06_final")
