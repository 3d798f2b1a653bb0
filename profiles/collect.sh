#!/bin/bash
# Collect the profiles of a round on the MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r01d'
# Writes under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards (see README.md).
set -u
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# 1. the bench line as the driver runs it (with the CPU baseline leg)
timeout -k 10 400 python3 bench.py > $out/bench_line.json 2> $out/bench_line.err
# 2. per-kernel times of the same workload (kernel trace + stats only)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace -- python3 bench.py --no-cpu-baseline > $out/bench_line_under_rocprof.json 2> $out/ktrace.err
cp $out/ktrace/*/*kernel_stats.csv $out/kernel_stats.csv
# 3. HBM traffic of the sketch kernel: one counter per pass, no tracing
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out/pmc_$ctr -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-solve > /dev/null 2> $out/pmc_$ctr.err
done
python3 - "$out" <<'P'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for f in glob.glob(f"{out}/pmc_{ctr}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    res[ctr] = {k: tot[k] / n[k] for k in tot}
json.dump(res, open(f"{out}/pmc_hbm_counters.json", "w"), indent=1)
for k in res.get("FETCH_SIZE", {}):
    if "phi_sketch_kernel<2" in k or "MODE_PROBE" in k:
        print(k[:60], "FETCH_SIZE kB", res["FETCH_SIZE"][k], "WRITE_SIZE kB", res["WRITE_SIZE"].get(k))
P
head -8 $out/kernel_stats.csv | cut -c1-140
cat $out/bench_line.json
