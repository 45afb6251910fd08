"""Per-kernel counter table of the step kernels from the passes of tools/step_pmc.sh.

usage: python tools/step_pmc_summary.py <gpurun_out/tag> <profiles/rNN_step_pmc.json> [steps-in-the-profiled-run]
Every counter is averaged per launch of a kernel; the verdict per kernel follows from the ratios:
  wait_frac    = SQ_WAIT_ANY / SQ_WAVE_CYCLES        waves parked at s_waitcnt / barriers (latency- or queue-bound)
  issue_frac   = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES waves issuing
  vmem_level   = SQ_INST_LEVEL_VMEM / SQ_WAVE_CYCLES vector-memory instructions in flight per wave
  occupancy    = SQ_WAVE_CYCLES / SQ_BUSY_CYCLES     resident waves per busy SQ cycle (SQ_BUSY_CYCLES is summed over SEs on gfx9)
  l2_hit       = TCC_HIT / (TCC_HIT + TCC_MISS)
  ta_busy      = TA_BUSY / (GRBM_GUI_ACTIVE * TA instances)  address pipe occupancy
FETCH_SIZE / WRITE_SIZE are KiB (x 1024 -> bytes), separate passes (MI355X_MICROARCH.md)."""
import collections
import csv
import json
import os
import sys


def per_kernel(path):
    f = os.path.join(path, "bench_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(f):
        return agg
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("magent_amd::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        agg[k]["_grid"].append(float(r["Grid_Size"]))
        agg[k]["_vgpr"].append(float(r["VGPR_Count"]))
    return agg


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {}
    for group in ("sq", "tcc", "tcc2", "ta", "fetch", "write"):
        for k, c in per_kernel(os.path.join(src, group)).items():
            o = out.setdefault(k, {})
            for name, v in c.items():
                if name.startswith("_"):
                    if group == "sq":
                        o[{"_ns": "avg_us_in_sq_pass", "_grid": "grid_threads", "_vgpr": "vgprs"}[name]] = round(sum(v) / len(v) / (1e3 if name == "_ns" else 1), 2)
                    continue
                o[name] = round(sum(v) / len(v), 1)
                o.setdefault("launches", len(v))
    for k, o in out.items():
        wc = o.get("SQ_WAVE_CYCLES")
        if wc:
            o["wait_frac"] = round(o.get("SQ_WAIT_ANY", 0) / wc, 3)
            o["issue_stall_frac"] = round(o.get("SQ_WAIT_INST_ANY", 0) / wc, 3)
            o["issue_frac"] = round(o.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3)
            o["vmem_in_flight_per_wave"] = round(o.get("SQ_INST_LEVEL_VMEM", 0) / wc, 3)
            if o.get("SQ_BUSY_CYCLES"):
                o["waves_per_busy_cycle"] = round(wc / o["SQ_BUSY_CYCLES"], 2)
        h, m = o.get("TCC_HIT_sum"), o.get("TCC_MISS_sum")
        if h is not None and m is not None and h + m > 0:
            o["l2_hit"] = round(h / (h + m), 3)
        if "FETCH_SIZE" in o:
            o["fetch_MB"] = round(o["FETCH_SIZE"] * 1024 / 1e6, 2)
        if "WRITE_SIZE" in o:
            o["write_MB"] = round(o["WRITE_SIZE"] * 1024 / 1e6, 2)
    keep = {k: v for k, v in out.items() if k.startswith("k_")}
    json.dump(keep, open(dst, "w"), indent=1, sort_keys=True)
    cols = ["launches", "avg_us_in_sq_pass", "SQ_WAVES", "wait_frac", "issue_frac", "vmem_in_flight_per_wave", "waves_per_busy_cycle", "l2_hit",
            "TCC_ATOMIC_sum", "fetch_MB", "write_MB"]
    print("| kernel | " + " | ".join(cols) + " |")
    print("|---|" + "---|" * len(cols))
    for k in sorted(keep, key=lambda k: -(keep[k].get("avg_us_in_sq_pass", 0) * keep[k].get("launches", 0))):
        print("| `%s` | " % k[:40] + " | ".join(str(keep[k].get(c, "")) for c in cols) + " |")


if __name__ == "__main__":
    main()
