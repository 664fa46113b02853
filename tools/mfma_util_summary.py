#!/usr/bin/env python3
"""Per-kernel matrix-pipe utilisation from ONE rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE: SQ and GRBM slots,
no tracing domains beside --kernel-trace), north_star's "MFMA utilisation against peak" as the hardware counts it.
util = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (sum(GRBM_GUI_ACTIVE) / 8 XCDs * 1024 SIMDs) per kernel key (the gfx94x MfmaUtil formula -
MI355X_MICROARCH.md, counters: ROCm 7.2 has no gfx950 derived-metric section - with GRBM_GUI_ACTIVE read as the SUM over the 8 XCDs'
counters: a 2.08 ms launch reads 29.0 M = 8 x 3.6 M cycles, i.e. 1.74 GHz, the power-capped clock).  It is the fraction of SHADER CYCLES
the matrix pipes were busy at the clock the kernel ran at; against the 2.4 GHz data-sheet peak multiply by clock / 2.4 GHz
(`clock_ghz` below = GUI_ACTIVE / 8 / the launch's duration is not available from this pass alone: see the kernel trace).  SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles
(16 per v_mfma_f32_16x16x32_f16), so busy / 16 * 16 384 FLOP is also the number of matrix FLOPs the hardware issued - printed
beside the util as `mfma_tflop_per_launch` for a cross-check against the algorithmic FLOPs of bench.py's roofline.
usage: python tools/mfma_util_summary.py <p_counter_collection.csv> [clips] [commit] > profiles/rNN_mfma_util.json"""
import collections, csv, json, sys
from traffic_summary import norm

busy, act, n = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    k = norm(r["Kernel_Name"])
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        busy[k] += float(r["Counter_Value"])
        n[k] += 1
    elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        act[k] += float(r["Counter_Value"])
out = {}
for k in sorted(busy):
    if busy[k] <= 0:
        continue
    out[k] = {"launches": n[k], "mfma_busy_cycles_per_launch": round(busy[k] / n[k]), "gui_active_per_launch": round(act[k] / n[k]),
              "mfma_util": round(busy[k] / (act[k] / 8.0 * 1024.0), 4) if act[k] else None,
              "mfma_tflop_per_launch": round(busy[k] / n[k] / 16.0 * 16384 / 1e12, 4)}
json.dump({"clips": int(sys.argv[2]) if len(sys.argv) > 2 else None, "commit": sys.argv[3] if len(sys.argv) > 3 else None,
           "note": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs), summed over a kernel's launches: busy fraction of shader cycles; "
                   "mfma_tflop_per_launch = busy / 16 cycles per 16x16x32 MFMA * 16 384 FLOP", "kernels": out}, sys.stdout, indent=1)
