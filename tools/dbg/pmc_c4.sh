#!/bin/bash
# C4 only, one stream: instruction-issue counters per kernel of the sorted pipeline (is a kernel VALU-issue bound?)
set -u
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/pmc_c4
mkdir -p $O
CFG="--configs ${1:-c4} --reps 5 --debug overlap=0"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $O/a -o s --output-format csv -- python tools/bench_configs.py $CFG > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --kernel-trace -d $O/b -o s --output-format csv -- python tools/bench_configs.py $CFG > $O/b.log 2>&1
tail -3 $O/a.log $O/b.log
python - <<'PY'
import csv, glob, collections
for run in ("a", "b"):
    f = glob.glob(f"gpurun_out/pmc_c4/{run}/**/*counter_collection.csv", recursive=True)
    if not f: print(run, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
    for k in sorted(acc):
        print(run, k[:40], "calls", n[k], {c: round(v / n[k]) for c, v in acc[k].items()})
PY
