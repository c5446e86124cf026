export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --pool 512 --tile 8 --from-raw 1"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc6 -o p6 -- $B > $R/gpurun_out/pmc6.log 2>&1; echo rc=$?
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SMEM TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/pmc7 -o p7 -- $B > $R/gpurun_out/pmc7.log 2>&1; echo rc=$?
