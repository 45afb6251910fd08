"""Turn rocprofv3 output directories (gpurun_out/...) into the small committed summaries under profiles/.

usage: python tools/summarize_profile.py <round-tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>]
  stats_dir : rocprofv3 --kernel-trace --stats --output-format csv   (bench_kernel_stats.csv)
  pmc_*_dir : rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE  (bench_counter_collection.csv), separate passes
Writes profiles/<tag>_kernel_stats.csv (verbatim), profiles/<tag>_summary.md and profiles/render_pmc.json.
FETCH_SIZE / WRITE_SIZE are in KiB (calibrated: k_paint writes exactly w*h*8 bytes and reports 7812.5 for 8e6 B).
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pmc_avg(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(path, "bench_counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    tag, stats_dir = sys.argv[1], sys.argv[2]
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    shutil.copy(os.path.join(stats_dir, "bench_kernel_stats.csv"), os.path.join(out_dir, tag + "_kernel_stats.csv"))
    rows = list(csv.DictReader(open(os.path.join(stats_dir, "bench_kernel_stats.csv"))))
    lines = ["# rocprofv3 summary %s" % tag, "",
             "command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-extras`",
             "", "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in rows[:24]:
        lines.append("| `%s` | %s | %.1f | %.3f | %s |" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                        float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    log = stats_dir.rstrip("/") + ".log"
    if os.path.exists(log):
        for line in open(log, errors="ignore"):
            if line.startswith("{") and '"roofline"' in line:
                rec = json.loads(line)
                lines += ["", "bench.py's own HIP-event timing in this very run (must agree with the k_render row above; the rocprof row also "
                          "counts the warm-up launches and the 5 breakdown steps after the timed region, whose groups are smaller):",
                          "", "```", json.dumps(rec["roofline"]), "```",
                          "", "whole job in the profiled run: %.3e %s, %.3f ms/step" % (rec["value"], rec["unit"], rec["ms_per_step"])]
    if len(sys.argv) >= 5:
        fetch, nf = pmc_avg(sys.argv[3], "FETCH_SIZE")
        write, _ = pmc_avg(sys.argv[4], "WRITE_SIZE")
        lines += ["", "## HBM-side traffic per launch (PMC, separate passes; KiB * 1024)", "",
                  "| kernel | launches | FETCH_SIZE MB | WRITE_SIZE MB |", "|---|---|---|---|"]
        for k in sorted(fetch, key=lambda k: -(fetch[k] + write.get(k, 0))):
            if not k.startswith("void magent_amd") and not k.startswith("magent_amd"):
                continue
            lines.append("| `%s` | %d | %.2f | %.2f |" % (k.split("(")[0][:70], nf[k], fetch[k] * 1024 / 1e6, write.get(k, 0) * 1024 / 1e6))
        rk = sorted((k for k in fetch if "k_render" in k), key=lambda k: -write.get(k, 0))   # the render of the bench workload: the one that writes most
        if rk:
            k = rk[0]
            # agents per k_render launch of the PMC runs (the populations shrink as agents die): mean of start / end, from the runs' own bench lines
            agents = []
            for d in (sys.argv[3], sys.argv[4]):
                lg = d.rstrip("/") + ".log"
                if os.path.exists(lg):
                    for line in open(lg, errors="ignore"):
                        if line.startswith("{") and '"agents_at_start"' in line:
                            c = json.loads(line)["config"]
                            agents.append((sum(c["agents_at_start"]) + sum(c["agents_at_end"])) / (2.0 * len(c["agents_at_start"])))
            per_launch = sum(agents) / len(agents) if agents else None
            rec = {"kernel": k.split("(")[0], "fetch_bytes_per_launch": fetch[k] * 1024, "write_bytes_per_launch": write.get(k, 0) * 1024,
                   "hbm_bytes_per_launch": (fetch[k] + write.get(k, 0)) * 1024,
                   "agents_per_launch": per_launch,
                   "hbm_bytes_per_agent": (fetch[k] + write.get(k, 0)) * 1024 / per_launch if per_launch else None,
                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB*1024; WRITE_SIZE calibrated on k_paint "
                           "(it writes exactly w*h*4 bytes packed / w*h*8 unpacked); FETCH_SIZE is uncalibrated for 8-byte gathers on gfx950 and counts "
                           "Infinity-Cache hits (MI355X_MICROARCH.md, HBM section) -- an upper bound on HBM reads",
                   "source": tag}
            # profiles/render_pmc.json holds one record per workload (bench.py scales `roofline.traffic` from it); this tool refreshes the headline's
            pmc_path = os.path.join(out_dir, "render_pmc.json")
            allrec = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
            if "workloads" not in allrec:
                allrec = {"note": rec["note"], "workloads": {}}
            rec.pop("note")
            rec["workload"] = "battle 1000x1000, 2x400k random (the headline)"
            allrec["workloads"]["battle_1000"] = rec
            json.dump(allrec, open(pmc_path, "w"), indent=1)
            lines += ["", "k_render: algorithmic bytes per launch = n_g * 4 * VH*VW*C; measured write %.3f GB, fetch %.3f GB" %
                      (rec["write_bytes_per_launch"] / 1e9, rec["fetch_bytes_per_launch"] / 1e9)]
    target = os.path.join(out_dir, tag + "_summary.md")
    notes = ""          # hand-written notes behind the generated tables survive a refresh
    if os.path.exists(target):
        old = open(target).read()
        k = old.find("\n## Round")
        notes = old[k:] if k >= 0 else ""
    open(target, "w").write("\n".join(lines) + "\n" + notes)
    print("\n".join(lines[:16]))


if __name__ == "__main__":
    main()
