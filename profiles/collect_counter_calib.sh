# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/hbm_counter_calib.hip); separate --pmc passes
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
[ -x $R/tools/hbm_counter_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $R/tools/hbm_counter_calib.hip -o $R/tools/hbm_counter_calib
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/calib_f -o c -- $R/tools/hbm_counter_calib > $R/gpurun_out/calib_f.log 2>&1; echo rc=$?
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/calib_w -o c -- $R/tools/hbm_counter_calib > $R/gpurun_out/calib_w.log 2>&1; echo rc=$?
python3 - <<'PY'
import csv, glob, os
R=os.environ["GRAFT_REPO_ROOT"]
for d in ("calib_f","calib_w"):
    for f in glob.glob(R+"/gpurun_out/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "calib_" in r["Kernel_Name"]:
                v=float(r["Counter_Value"])
                print("%-14s %-10s %14.0f KB-units = %.4f x of 1 GiB (as KB x 1024)" % (r["Kernel_Name"].split("(")[0], r["Counter_Name"], v, v*1024/2**30))
PY
