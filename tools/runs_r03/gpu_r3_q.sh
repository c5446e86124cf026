#!/bin/bash
# call 21: instruction counters of the shipped chain kernel (two reads per wave) and of the one-read kernel, same batch
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O; cd /tmp
W="python $R/tools/pmc_workload.py --reads 0 --ea-reads 8192 --reps 2"
timeout 120 $W > $O/units.json 2> $O/units.err; echo "units rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq1 -o sq1 -- $W > $O/sq1.log 2>&1; echo "sq1 rc=$?"
python3 - <<PY
import csv, glob, collections
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob("$O/sq1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    print(k, len(n[k]), {c: "%.4g" % (x / len(n[k])) for c, x in v.items()})
PY
cat $O/units.json
