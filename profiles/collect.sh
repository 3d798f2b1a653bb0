#!/bin/bash
# Collect the profiles of a round on the MI355X box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r02a'
# Writes under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards (see README.md).
set -u
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
# 1. the bench line as the driver runs it (with the CPU baseline and the extra legs)
timeout -k 10 500 python3 $B > $out/bench_line.json 2> $out/bench_line.err
# 2. per-kernel times of the same workload (kernel trace + stats only)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace -- python3 $B --no-cpu-baseline --no-extra-legs --job-repeats 0 > $out/bench_line_under_rocprof.json 2> $out/ktrace.err
cp $out/ktrace/*/*kernel_stats.csv $out/kernel_stats.csv
# 3. HBM traffic of the sketch kernel: one counter per pass, no tracing
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out/pmc_$ctr -- python3 $B --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs --no-solve --job-repeats 0 > /dev/null 2> $out/pmc_$ctr.err
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
    if "phi_sketch_kernel<2" in k or "phi_sketch_pool_kernel" in k:
        print(k[:60], "FETCH_SIZE kB", res["FETCH_SIZE"][k], "WRITE_SIZE kB", res["WRITE_SIZE"].get(k))
P
# 3b. the same three collections for the larger read sets (C3: 10x short reads, C4: long noisy reads, C5s: 200 walks, 30x):
#     per-kernel times, HBM counters, and the SQ set (instruction counts, wave cycles, waits) of the sketch kernel
for cfg in C3 C4 C5s; do
  st=5; [ $cfg = C5s ] && st=3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktrace_$cfg -- python3 $B --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs --job-repeats 0 > $out/bench_line_${cfg}_under_rocprof.json 2> $out/ktrace_$cfg.err
  cp $out/ktrace_$cfg/*/*kernel_stats.csv $out/kernel_stats_$cfg.csv
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"; do
    tagc=$(echo $ctr | cut -d" " -f1)
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $out/pmc_${cfg}_$tagc -- python3 $B --config $cfg --steps $st --warmup 1 --no-cpu-baseline --no-extra-legs --no-solve --job-repeats 0 > /dev/null 2> $out/pmc_${cfg}_$tagc.err
  done
done
python3 - "$out" <<'P'
import csv, glob, json, sys, collections
out = sys.argv[1]
res = {}
for cfg in ("C3", "C4", "C5s"):
    per = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/pmc_{cfg}_*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "phi_sketch_kernel<2" in r["Kernel_Name"] or "phi_sketch_kernel<(int)2" in r["Kernel_Name"] or "phi_sketch_pool_kernel" in r["Kernel_Name"]:
                k = per[r["Counter_Name"]]; k[0] += float(r["Counter_Value"]); k[1] += 1
    res[cfg] = {c: v[0] / max(1, v[1]) for c, v in per.items()}
    res[cfg]["launches_averaged"] = {c: v[1] for c, v in per.items()}
json.dump(res, open(f"{out}/sketch_kernel_counters_C3_C4_C5s.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
P
# 4. the other configurations (bench lines only)
for cfg in C3 C4 C1syn C2r C3r; do
  timeout -k 10 300 python3 $B --config $cfg --steps 100 --warmup 10 --no-cpu-baseline --no-extra-legs --job-repeats 3 > $out/bench_line_$cfg.json 2> $out/bench_line_$cfg.err
done
timeout -k 10 300 python3 $B --config C5s --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs > $out/bench_line_C5s.json 2> $out/bench_line_C5s.err
PHI_TIMING=1 timeout -k 10 600 python3 $B --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > $out/bench_line_C5.json 2> $out/bench_line_C5.err
timeout -k 10 600 python3 $B --config C5 --scaling strong --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-solve > $out/bench_line_C5_strong.json 2> $out/bench_line_C5_strong.err
head -8 $out/kernel_stats.csv | cut -c1-140
cat $out/bench_line.json
# 5. config 5 at its stated size from FILES (20 GB on the RAM disk): process start -> closed FASTA with its stage table
timeout -k 10 900 python3 $GRAFT_REPO_ROOT/profiles/c5_files.py --config C5 --runs 5 --out $out/c5_files.json > $out/c5_files.log 2>&1
tail -2 $out/c5_files.log | cut -c1-600
