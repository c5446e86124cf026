export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --pool 512 --tile 16 --from-raw 1"
timeout 250 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/rawpmc_f -o p -- $B > /dev/null 2>&1
timeout 250 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/rawpmc_w -o p -- $B > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
tot=collections.defaultdict(float); n=collections.defaultdict(set)
for f in glob.glob(R+"/gpurun_out/rawpmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-40:]
        tot[(k,r["Counter_Name"])]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])].add(r["Dispatch_Id"])
for k in sorted(tot):
    if "np_ed" in k[0] or "mom" in k[0]:
        corr = 2.0 if k[1]=="FETCH_SIZE" else 1.0
        print("%-44s %-11s %8.1f MB per launch (8192 reads)" % (k[0], k[1], tot[k]/len(n[k])*1024*corr/1e6))
PY
