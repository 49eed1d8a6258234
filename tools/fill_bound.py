#!/usr/bin/env python3
"""Per-op operand-staging ("fill") bound of a forward, from a per-op table of tools/gpu_explore.py (profiles/*_per_op.txt).

Every MFMA conv kernel stages both operands from L2 / Infinity Cache into LDS; a CU pulls about 12.5 B/clk that way (30 GB/s at
2.4 GHz: DESIGN.md section 7, docs/experiments.md section 4), 7.7 TB/s over 256 CUs.  Bytes a configuration stages for a layer:
  implicit GEMM, tile TM x TN: M N K 2 (1 / TM + 1 / TN) with K = k^2 Cin;
  patch kernel (3x3 s1), 128-pixel tiles x TN couts: pixels 180 / 128 entries of Cin per cout tile, weights 9 Cin TN per tile.
Prints time, staged MB and the ratio time / (staged bytes / 7.7 TB/s) per conv op, and the totals.   usage: fill_bound.py <per_op.txt> [batch]
"""
import re
import sys

TILES = {0: (256, 32), 1: (256, 64), 2: (128, 128), 3: (128, 64), 4: (256, 128), 5: (64, 128), 6: (64, 64), 7: (256, 256), 8: (128, 128), 9: (128, 64),
         14: (128, 128), 15: (128, 64)}
PATCH = {10: 128, 11: 64, 12: 128, 13: 64}
path, B = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64
RATE = 256 * 12.5 * 2.4e9  # bytes / s
tot_t = tot_b = 0.0
rows = []
for line in open(path):
    m = re.match(r"\s*([\d.]+) ms kind 1 (\S+)\s+(\d)x\d+s(\d)\s+(\d+)->\s*(\d+) @(\d+)x(\d+).*cfg (0x[0-9a-f]+)", line)
    if not m:
        continue
    t, name, k, s, cin, cout, oh, ow, cfg = float(m[1]), m[2], int(m[3]), int(m[4]), int(m[5]), int(m[6]), int(m[7]), int(m[8]), int(m[9], 16)
    M, K, tile = B * oh * ow, k * k * cin, cfg >> 4
    if tile in PATCH:
        tn = PATCH[tile]
        ntn = -(-cout // tn)
        staged = 2.0 * (M / 128) * ntn * (180 * cin + 9 * cin * tn)
    else:
        tm, tn = TILES[tile]
        if tile >= 14:  # weights-resident: pixels only, once per cout tile
            staged = 2.0 * M * K * -(-cout // tn)
        else:
            staged = 2.0 * M * max(cout, tn) * K * (1.0 / tm + 1.0 / tn)
    fb = staged / RATE * 1e3
    rows.append((t, name, staged / 1e6, fb))
    tot_t += t
    tot_b += staged
print(f"{'op':28s} {'ms':>8s} {'staged MB':>10s} {'fill-bound ms':>14s} {'time / bound':>13s}")
for t, name, mb, fb in rows:
    print(f"{name:28s} {t:8.4f} {mb:10.1f} {fb:14.4f} {t / fb:13.2f}")
print(f"{'plain conv ops, total':28s} {tot_t:8.3f} {tot_b / 1e6:10.0f} {tot_b / RATE * 1e3:14.3f} {tot_t / (tot_b / RATE * 1e3):13.2f}")
