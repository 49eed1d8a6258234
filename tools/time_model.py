#!/usr/bin/env python3
"""The time model of DESIGN.md section 7 held against a per-op table of tools/gpu_explore.py (profiles/*_per_op.txt), conv op by conv op.

    T = FIXED + staged bytes per CU / STAGE_RATE + stored bytes per CU / STORE_RATE      (one workgroup per CU: the terms do not overlap)
    T = FIXED + max(staged, stored term)                                                 (two or more workgroups per CU overlap them)

FIXED = 7 us (launch, prologue, SiLU + staging), STAGE_RATE = 40 GB/s per CU (LDS-DMA operand stream), STORE_RATE = 20 GB/s per CU -- the
three numbers the chain kernel's ablations measured (docs/experiments.md section 0.5).  Bytes a configuration stages for a layer:
  implicit GEMM / persistent, tile TM x TN: per tile K (TM + TN) 2 bytes; tiles = ceil(M / TM) ceil(Cout / TN);
  patch kernel (3x3 s1), 128-pixel tiles x TN couts: per tile (180 Cin + 9 Cin TN) 2 bytes;
  weights-resident 1x1: pixels only.
A launch's tiles run in rounds of 256 x (workgroups per CU); the time of a round is that of its busiest CU.  Floor: the layer's
algorithmic HBM bytes (input + output) at HBM_RATE = 4.5 TB/s (the 160 x 160 / 80 x 80 1x1 layers).
usage: time_model.py <per_op.txt> [batch]
"""
import math
import re
import sys

TILES = {0: (256, 32, 2), 1: (256, 64, 2), 2: (128, 128, 2), 3: (128, 64, 3), 4: (256, 128, 1), 5: (64, 128, 3), 6: (64, 64, 4), 7: (256, 256, 1), 8: (128, 128, 2), 9: (128, 64, 2),
         14: (128, 128, 2), 15: (128, 64, 2)}  # TM, TN, workgroups per CU
PATCH = {10: (128, 2), 11: (64, 3), 12: (128, 2), 13: (64, 3)}
FIXED, STAGE_RATE, STORE_RATE, NCU, HBM_RATE = 7.0, 40e3, 20e3, 256, 4.5e6  # us, bytes / us per CU, bytes / us
path, B = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64
rows, tot_t, tot_m = [], 0.0, 0.0
for line in open(path):
    m = re.match(r"\s*([\d.]+) ms kind 1 (\S+)\s+(\d)x\d+s(\d)\s+(\d+)->\s*(\d+) @(\d+)x(\d+).*cfg (0x[0-9a-f]+)", line)
    if not m:
        continue
    t, name, k, s, cin, cout, oh, ow, cfg = float(m[1]) * 1e3, m[2], int(m[3]), int(m[4]), int(m[5]), int(m[6]), int(m[7]), int(m[8]), int(m[9], 16)
    M, K, tile = B * oh * ow, k * k * cin, cfg >> 4
    if tile in PATCH:
        tn, wpc = PATCH[tile]
        tiles = math.ceil(M / 128) * math.ceil(cout / tn)
        staged_wg, stored_wg = 2.0 * (180 * cin + 9 * cin * tn), 2.0 * 128 * min(tn, cout)
    else:
        tm, tn, wpc = TILES[tile]
        tiles = math.ceil(M / tm) * math.ceil(cout / tn)
        staged_wg = 2.0 * K * (tm if tile >= 14 else tm + tn)
        stored_wg = 2.0 * tm * min(tn, cout)
    per_cu = tiles / NCU  # tiles of the busiest CU, as a continuous quantity for multi-workgroup kernels ...
    if wpc == 1:          # ... and whole rounds where a CU holds one workgroup at a time
        per_cu = math.ceil(tiles / NCU)
        model = per_cu * (FIXED + staged_wg / STAGE_RATE + stored_wg / STORE_RATE)
    else:
        per_cu = max(per_cu, 1.0)
        model = FIXED + per_cu * max(staged_wg / STAGE_RATE, stored_wg / STORE_RATE)
    hbm = 2.0 * (M * s * s * cin + M * cout) / HBM_RATE
    model = max(model, hbm)
    rows.append((name, f"{k}x{k}s{s} {cin}->{cout} @{oh}", cfg, tiles, wpc, t, model))
    tot_t += t
    tot_m += model
print(f"# {path}: measured per-op time of the plain conv launches against FIXED {FIXED} us + staged / {STAGE_RATE / 1e3:.0f} GB/s + stored / {STORE_RATE / 1e3:.0f} GB/s per CU")
print(f"{'op':28s} {'shape':26s} {'cfg':>5s} {'tiles':>6s} {'wg/CU':>5s} {'measured us':>12s} {'model us':>9s} {'measured / model':>17s}")
for name, shape, cfg, tiles, wpc, t, model in rows:
    print(f"{name:28s} {shape:26s} {cfg:#5x} {tiles:6d} {wpc:5d} {t:12.1f} {model:9.1f} {t / model:17.2f}")
print(f"{'plain conv ops, total':28s} {'':26s} {'':5s} {'':6s} {'':5s} {tot_t:12.0f} {tot_m:9.0f} {tot_t / tot_m:17.2f}")
