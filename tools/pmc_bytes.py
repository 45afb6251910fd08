"""Per-kernel HBM-side traffic from the separate `rocprofv3 --pmc` passes of tools/measure.sh's `bytes` job.

usage: python tools/pmc_bytes.py <gpurun_out/tag> <out.json>
FETCH_SIZE / WRITE_SIZE are KiB (x 1024 -> bytes; MI355X_MICROARCH.md, HBM / rocprofv3 section: separate passes; WRITE_SIZE is exact
for whole-line stores -- calibrated on k_paint in round 2 -- FETCH_SIZE counts Infinity-Cache hits too: an upper bound on HBM reads);
TCC_EA0_RDREQ / WRREQ are the L2's requests to the fabric (32 or 64 bytes each).  Every figure is the average per launch."""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter):
    f = os.path.join(path, counter, "bench_counter_collection.csv")
    agg = collections.defaultdict(list)
    ns = collections.defaultdict(list)
    if os.path.exists(f):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("magent_amd::", "")
            agg[k].append(float(r["Counter_Value"]))
            ns[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return agg, ns


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {}
    for counter, key, scale in (("FETCH_SIZE", "fetch_bytes", 1024.0), ("WRITE_SIZE", "write_bytes", 1024.0),
                                ("TCC_EA0_RDREQ_sum", "read_requests", 1.0), ("TCC_EA0_WRREQ_sum", "write_requests", 1.0)):
        agg, ns = per_kernel(src, counter)
        for k, v in agg.items():
            o = out.setdefault(k, {})
            o[key] = sum(v) / len(v) * scale
            o["launches"] = len(v)
            o.setdefault("avg_us", round(sum(ns[k]) / len(ns[k]) / 1e3, 2))
    keep = {k: v for k, v in out.items() if k.startswith("k_")}
    json.dump(keep, open(dst, "w"), indent=1, sort_keys=True)
    print("%-34s %6s %9s %10s %10s %10s %10s" % ("kernel", "calls", "us", "fetch MB", "write MB", "rd req", "wr req"))
    for k in sorted(keep, key=lambda k: -(keep[k].get("fetch_bytes", 0) + keep[k].get("write_bytes", 0)) * keep[k]["launches"]):
        o = keep[k]
        print("%-34s %6d %9.1f %10.2f %10.2f %10.0f %10.0f" % (k[:34], o["launches"], o.get("avg_us", 0), o.get("fetch_bytes", 0) / 1e6,
                                                                 o.get("write_bytes", 0) / 1e6, o.get("read_requests", 0), o.get("write_requests", 0)))


if __name__ == "__main__":
    main()
