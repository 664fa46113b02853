#!/usr/bin/env python3
"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at 64 B, i.e. reads exactly half of a
wide coalesced read stream -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Both counters are in KiB."""
import collections, csv, json, re, sys


def norm(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"tapgemm_kernel<Elem(\w+), (\d+), (\d+), \d+, \d+, (\d+), \d+, (\d+), \w+>", name)
    if m:
        return f"tapgemm<{m.group(1).lower()},{m.group(2)}x{m.group(3)},mode{m.group(4)},e{m.group(5)}>"
    m = re.match(r"phasegemm_kernel<Elem(\w+), (\d+), (\d+)>", name)
    if m:  # csrc/phasegemm_kernel.h: the profiler names it by its tile, 256x256
        return f"tapgemm<{m.group(1).lower()},256x256,mode{m.group(2)},e{m.group(3)}>"
    m = re.match(r"patchconv64_kernel<Elem(\w+), (\d+), (\d+), (\d+)>", name)
    if m:  # same key as ops.py's profiler uses for the patch kernel (tile code 999x64 / 999x128)
        return f"tapgemm<{m.group(1).lower()},999x{m.group(4)},mode{m.group(2)},e{m.group(3)}>"
    m = re.match(r"resblock_kernel<Elem\w+, (\d+), (\d+)>", name)
    if m:
        return f"l2s_resblock_fused<C{m.group(1)},k{m.group(2)}>"
    m = re.match(r"resstage_kernel<Elem\w+, (\d+)>", name)
    if m:  # csrc/resblock.hip: the three ResBlocks of a narrow stage in one launch
        return f"l2s_resstage_fused<C{m.group(1)}>"
    if name.startswith("stem_pool_kernel"):
        return "l2s_stem_pool_fused"
    if name.startswith("basicblock_kernel"):   # csrc/basicblock.hip: the path launches it once per layer (l2s_basiclayer_fused)
        return "l2s_basiclayer_fused"
    m = re.match(r"respair_phase_kernel<Elem\w+, (\d+), (\d+)>", name)
    if m:  # csrc/respair_phase.hip: the phase-staggered pair kernel (C = 256 / 128 / 64), same keys as respair.hip's
        return f"l2s_respair<C{m.group(1)},{'last' if m.group(2) == '1' else 'mid'}>"
    m = re.match(r"respair_kernel<Elem\w+, (\d+), (\d+)>", name)
    if m:  # csrc/respair.hip: KIND 0 = mid pair, 1 = last pair of a ResBlock (k is a runtime argument)
        return f"l2s_respair<C{m.group(1)},{'last' if m.group(2) == '1' else 'mid'}>"
    if name.startswith("attention_resident_kernel"):
        return "l2s_attention"
    if name.startswith("layernorm_rows_kernel"):
        return "l2s_layernorm"
    m = re.match(r"(\w+)_kernel", name)
    return "l2s_" + m.group(1) if m else name[:60]


def load(path):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        k = norm(r["Kernel_Name"])
        agg[k][0] += float(r["Counter_Value"])
        agg[k][1] += 1
    return agg


def main():
    f, w = load(sys.argv[1]), load(sys.argv[2])
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else None
    out = {}
    for k in sorted(set(f) | set(w)):
        fk = f[k][0] / max(f[k][1], 1) if k in f else 0.0
        wk = w[k][0] / max(w[k][1], 1) if k in w else 0.0
        out[k] = {"launches": f[k][1] if k in f else w[k][1], "fetch_kib_raw_per_launch": round(fk, 1),
                  "write_kib_per_launch": round(wk, 1), "hbm_bytes_per_launch": round((2.0 * fk + wk) * 1024)}
    commit = sys.argv[4] if len(sys.argv) > 4 else None
    json.dump({"batch": batch, "frames": 100, "commit": commit, "note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch (gfx950 FETCH_SIZE halving corrected)",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
