"""Derive profiles/rNN_traffic.json (read by bench.py: the newest of r05 / r04 / r03) from the PMC passes of tools/collect_profiles.sh.

usage: python tools/make_traffic_json.py <dir with pmc_*_by_kernel.txt and pmc_*_fm_main.csv> <clouds per launch> [out.json] [commit]
  traffic of the dominant kernel (final FPS: fl_main_kernel, fm_main_kernel until round 2): mean over the bench's launches (the dispatches
      with the large counter values; the small ones are the input thinning) of FETCH_SIZE x 2 (gfx950: 128-byte
      requests are tallied at 64 B for 16 B/lane coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE,
      counter unit KiB;
  others_per_step: the same sums per kernel over the profiled steps (bench.py --steps 2 --warmup 1 = 3 steps);
  mfma: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) per kernel -- the fraction of SIMD
      cycles the matrix pipe was busy while the kernel ran."""
import csv
import json
import re
import sys

d, clouds = sys.argv[1], int(sys.argv[2])
STEPS = 3.0


def per_dispatch(path):
    rows = list(csv.DictReader(open(path)))
    vals = [float(r["Counter_Value"]) for r in rows]
    big = [v for v in vals if v > 0.25 * max(vals)]
    return sum(big) / len(big), len(big)


def by_kernel(path):
    out = {}
    for line in open(path):
        m = re.match(r"(.{60})\s+n=(\d+)\s+(.*)", line)
        if not m:
            continue
        name = m.group(1).strip()
        out[name] = dict(n=int(m.group(2)), **{k: float(v) for k, v in (kv.split("=") for kv in m.group(3).split())})
    return out


fetch, nl = per_dispatch(d + "/pmc_FETCH_SIZE_fm_main.csv")
write, _ = per_dispatch(d + "/pmc_WRITE_SIZE_fm_main.csv")
F, Wr = by_kernel(d + "/pmc_FETCH_SIZE_by_kernel.txt"), by_kernel(d + "/pmc_WRITE_SIZE_by_kernel.txt")
M, G = by_kernel(d + "/pmc_SQ_MFMA_by_kernel.txt"), by_kernel(d + "/pmc_GRBM_by_kernel.txt")
V = by_kernel(d + "/pmc_SQ_VALU_by_kernel.txt")


def find(tab, key):
    hits = [v for k, v in tab.items() if key in k]
    if not hits:
        return None
    agg = {}
    for h in hits:
        for k, v in h.items():
            agg[k] = agg.get(k, 0.0) + v
    return agg


others, mfma = {}, {}
for key in ("dec_fused", "knn_graph_key_kernel", "knn_graph_slab_kernel", "knn_slab_order_kernel", "regress_tail_kernel", "regress_tail_sb_kernel", "linear_small_kernel", "linear_wide_kernel", "linear_wide_sb_kernel",
            "linear_lift_kernel", "skip_", "rl_main_kernel", "knn_insert_kernel", "knn_select_kernel", "knn_dup_lds"):
    f, w = find(F, key), find(Wr, key)
    if f and w:
        others[key] = {"launches_per_step": f["n"] / STEPS, "fetch_size_kib_per_step": f["FETCH_SIZE"] / STEPS,
                       "write_size_kib_per_step": w["WRITE_SIZE"] / STEPS,
                       "traffic_bytes_per_step": (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024 / STEPS}
    m, g, v = find(M, key), find(G, key), find(V, key)
    if m and g:
        simd_cycles = g["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        mfma[key] = {"mfma_busy_frac": m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                     "SQ_VALU_MFMA_BUSY_CYCLES": m["SQ_VALU_MFMA_BUSY_CYCLES"], "SQ_INSTS_MFMA": m["SQ_INSTS_MFMA"],
                     "SQ_INSTS_VALU_MFMA_MOPS_F32": m["SQ_INSTS_VALU_MFMA_MOPS_F32"],
                     "SQ_BUSY_CYCLES": m["SQ_BUSY_CYCLES"], "GRBM_GUI_ACTIVE": g["GRBM_GUI_ACTIVE"],
                     "SQ_INSTS_VALU": None if not v else v.get("SQ_INSTS_VALU")}
out = {
    "kernel": "fl_main_kernel<false> (final FPS; fm_main_kernel<8,1,false,false> until round 2)", "clouds_per_launch": clouds,
    "launches_averaged": nl,
    "fetch_size_kib": fetch, "write_size_kib": write,
    "fetch_correction": "x2 (gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane coalesced reads, "
                        "MI355X_MICROARCH.md HBM section)",
    "traffic_bytes_per_launch": (2 * fetch + write) * 1024,
    "source": "tools/collect_profiles.sh: rocprofv3 --pmc <one counter set> --kernel-trace -- python bench.py "
              "--no_cpu_baseline --no_extras --no_overlap --net_streams 1 --steps 2 --warmup 1 (one run per set)",
    "others_per_step": others,
    "mfma_utilisation": mfma,
    "mfma_note": "busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); "
                 "v_mfma_f32_16x16x4_f32 counts 32 busy cycles, v_mfma_f32_4x4x1 8",
}
import subprocess, datetime
out["provenance"] = {"commit": subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
                               or (sys.argv[4] if len(sys.argv) > 4 else "unknown"),
                     "date": datetime.date.today().isoformat(), "collected_by": "tools/collect_profiles.sh on a gpurun MI355X box"}
json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 else "profiles/r03_traffic.json", "w"), indent=1)
print(json.dumps({"traffic_GB_per_launch": out["traffic_bytes_per_launch"] / 1e9,
                  "mfma": {k: round(v["mfma_busy_frac"], 3) for k, v in mfma.items()},
                  "others_GB_per_step": {k: round(v["traffic_bytes_per_step"] / 1e9, 1) for k, v in others.items()}}, indent=1))
