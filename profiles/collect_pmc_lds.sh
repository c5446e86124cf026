export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --pool 512 --tile 16"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc5 -o p5 -- $B > $R/gpurun_out/pmc5.log 2>&1; echo rc5=$?
tail -2 $R/gpurun_out/pmc5.log
