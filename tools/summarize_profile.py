"""Turn the outputs of tools/profile_bench.sh <tag> into the committed summaries profiles/r<round>_<tag>_*.
    python tools/summarize_profile.py <tag> <round, e.g. 01>"""
import csv
import glob
import re
import shutil
import subprocess
import sys

tag, rnd = sys.argv[1], sys.argv[2]
pre = f"profiles/r{rnd}_{tag}_bench"
import os
f = max(glob.glob(f"gpurun_out/prof_{tag}/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)  # newest: gpurun_out keeps earlier runs
rows = list(csv.DictReader(open(f)))
with open(pre + "_kernel_stats.csv", "w") as o:
    w = csv.writer(o, quoting=csv.QUOTE_ALL)
    w.writerow(rows[0].keys())
    for r in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        r["Name"] = name.split("(scan::")[0] if "gae_fused_kernel" in name else name
        w.writerow(r.values())
subprocess.check_call([sys.executable, "tools/summarize_pmc.py", f"gpurun_out/pmc_fetch_{tag}", f"gpurun_out/pmc_write_{tag}", pre + "_pmc_hbm_traffic.json"])
shutil.copy(f"gpurun_out/bench_prof_{tag}.json", pre + "_under_rocprof.json")
shutil.copy(f"gpurun_out/bench_{tag}.json", pre + ".json")
print(open(pre + "_kernel_stats.csv").read()[:400])

# the cold (beyond-Infinity-Cache) leg alone, when tools/profile_bench.sh produced it
cold = glob.glob(f"gpurun_out/prof_cold_{tag}/**/*_kernel_stats.csv", recursive=True)
if cold:
    f = max(cold, key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    with open(pre + "_cold_kernel_stats.csv", "w") as o:
        w = csv.writer(o, quoting=csv.QUOTE_ALL)
        w.writerow(rows[0].keys())
        for r in rows:
            name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
            r["Name"] = name.split("(scan::")[0] if "gae_fused_kernel" in name else name
            w.writerow(r.values())
    subprocess.check_call([sys.executable, "tools/summarize_pmc.py", f"gpurun_out/pmc_fetch_cold_{tag}", f"gpurun_out/pmc_write_cold_{tag}",
                           pre + "_cold_pmc_hbm_traffic.json"])
    shutil.copy(f"gpurun_out/bench_prof_cold_{tag}.json", pre + "_cold_under_rocprof.json")
# the driver's invocation (--gpus 1 --steps 20 --warmup 5), three times
runs = sorted(glob.glob(f"gpurun_out/bench_driver*_{tag}.json"))
if runs:
    with open(pre + "_driver_style_runs.jsonl", "w") as o:
        for r in runs:
            o.write(open(r).read().strip() + "\n")
