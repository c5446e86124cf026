export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --pool 512 --tile 16"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/pmc1 -o p1 -- $B > $R/gpurun_out/pmc1.log 2>&1; echo rc1=$?
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc2 -o p2 -- $B > $R/gpurun_out/pmc2.log 2>&1; echo rc2=$?
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3 -o p3 -- $B > $R/gpurun_out/pmc3.log 2>&1; echo rc3=$?
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc4 -o p4 -- $B > $R/gpurun_out/pmc4.log 2>&1; echo rc4=$?
ls -R $R/gpurun_out/pmc1 | head; tail -3 $R/gpurun_out/pmc1.log
