#!/usr/bin/env python3
"""Compact per-kernel table from two rocprofv3 --pmc passes over tools/one_conv.py (see DESIGN.md / profiles)."""
import csv, sys
from collections import defaultdict
rows = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "conv_mfma" not in n and "conv_first" not in n:
            continue
        key = (n.split("(")[0].replace("void ", ""), int(r["Grid_Size"]) // 256)
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        rows[key]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key, cs in rows.items():
    a = defaultdict(float, {c: sum(v) / len(v) for c, v in cs.items()})
    wc, w = max(a["SQ_WAVE_CYCLES"], 1), max(a["SQ_WAVES"], 1)
    print(key, "blocks")
    print(f"   dur {a['dur_us']:.1f}us waves {w:.0f}  wave_cycles/wave {wc / w:.0f}  busy_cycles {a['SQ_BUSY_CYCLES']:.0f}")
    print(f"   frac of wave-cycles: WAIT_ANY {a['SQ_WAIT_ANY'] / wc:.2f}  WAIT_INST_ANY {a['SQ_WAIT_INST_ANY'] / wc:.2f} "
          f"ACTIVE_INST_ANY {a['SQ_ACTIVE_INST_ANY'] / wc:.2f} ACTIVE_VALU {a['SQ_ACTIVE_INST_VALU'] / wc:.2f} ACTIVE_LDS {a['SQ_ACTIVE_INST_LDS'] / wc:.2f}")
    print(f"   per wave: VALU {a['SQ_INSTS_VALU'] / w:.0f} SALU {a['SQ_INSTS_SALU'] / w:.0f} MFMA {a['SQ_INSTS_MFMA'] / w:.0f} "
          f"LDS {a['SQ_INSTS_LDS'] / w:.0f} VMEM {a['SQ_INSTS_VMEM'] / w:.0f}  mfma_busy {a['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} "
          f"lds_conflict {a['SQ_LDS_BANK_CONFLICT']:.0f} lds_active {a['SQ_LDS_IDX_ACTIVE']:.0f}")
