"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py into profiles/<tag>_pmc_hbm_traffic.json.
    python tools/summarize_pmc.py <fetch_dir> <write_dir> <out.json> [kernel-substring]
Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): rocprofv3 reports both counters in
KiB; FETCH_SIZE counts wide coalesced reads at half their bytes on gfx950 -> x2."""
import csv
import glob
import json
import os
import sys

fetch_dir, write_dir, out = sys.argv[1:4]
needle = sys.argv[4] if len(sys.argv) > 4 else "gae_fused_kernel"


def collect(d, counter):
    vals = []
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    for f in ([max(files, key=os.path.getmtime)] if files else []):   # newest run only: gpurun_out keeps earlier ones
        for row in csv.DictReader(open(f)):
            if needle in row["Kernel_Name"] and row["Counter_Name"] == counter:
                vals.append(float(row["Counter_Value"]))
    return vals


f, w = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
mean = lambda v: sum(v) / len(v)
alg = 65536 * 32 * 22
read_b, write_b = mean(f) * 1024 * 2, mean(w) * 1024
res = {
    "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline (one pass per counter)",
    "kernel": needle,
    "note": "rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their bytes (MI355X_MICROARCH.md, HBM) -> x2",
    "FETCH_SIZE": {"launches": len(f), "mean_KiB": mean(f), "min_KiB": min(f), "max_KiB": max(f)},
    "WRITE_SIZE": {"launches": len(w), "mean_KiB": mean(w), "min_KiB": min(w), "max_KiB": max(w)},
    "hbm_bytes_per_launch": {"read_corrected": read_b, "write": write_b, "total": read_b + write_b, "algorithmic": alg,
                             "ratio": (read_b + write_b) / alg},
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["hbm_bytes_per_launch"]))
