# PMC pass over tests/bench_eventalign.py (counters only: no trace domains beyond --kernel-trace, see the gpurun rules)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/tests/bench_eventalign.py --steps 1 --warmup 1 --cpu-sample 0 --pool 256 --tile 32"
timeout 250 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/eapmc1 -o p1 -- $B > $R/gpurun_out/eapmc1.log 2>&1; echo rc1=$?
timeout 250 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/eapmc2 -o p2 -- $B > $R/gpurun_out/eapmc2.log 2>&1; echo rc2=$?
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
tot=collections.defaultdict(float); n=collections.defaultdict(set)
for f in glob.glob(R+"/gpurun_out/eapmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "eventalign_chain" in r["Kernel_Name"]:
            tot[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot): print(k, tot[k]/max(1,len(n[k])), "per launch over", len(n[k]), "launches")
PY
