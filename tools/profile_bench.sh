#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box from the repo root): kernel trace + stats, then one PMC pass per
# counter (FETCH_SIZE, WRITE_SIZE: never combined with trace domains), then a plain bench run.  The profiled passes skip the
# cold leg (--no-cold: its launches carry the same kernel name and would mix DRAM-bound launches into the warm kernel's average).  Outputs under
# gpurun_out/<tag>_*; summarise with tools/summarize_profile.py <tag> and copy the summaries into profiles/.
#   bash tools/profile_bench.sh f
set -e
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/prof_$tag gpurun_out/pmc_fetch_$tag gpurun_out/pmc_write_$tag
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline --no-cold > gpurun_out/bench_prof_$tag.json 2> gpurun_out/prof_$tag.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold > /dev/null 2> gpurun_out/pmc_fetch_$tag.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cold > /dev/null 2> gpurun_out/pmc_write_$tag.err
# the cold (beyond-Infinity-Cache) leg alone: kernel statistics, then the two PMC passes
rm -rf gpurun_out/prof_cold_$tag gpurun_out/pmc_fetch_cold_$tag gpurun_out/pmc_write_cold_$tag
timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cold_$tag -- python3 bench.py --cold-only --steps 2000 --warmup 200 > gpurun_out/bench_prof_cold_$tag.json 2> gpurun_out/prof_cold_$tag.err
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_cold_$tag -- python3 bench.py --cold-only --steps 200 --warmup 50 > /dev/null 2> gpurun_out/pmc_fetch_cold_$tag.err
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_cold_$tag -- python3 bench.py --cold-only --steps 200 --warmup 50 > /dev/null 2> gpurun_out/pmc_write_cold_$tag.err
timeout -k 10 200 python3 bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
for k in 1 2 3; do timeout -k 10 100 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_driver${k}_$tag.json 2>> gpurun_out/bench_$tag.err; done
cut -c1-200 gpurun_out/bench_$tag.json
