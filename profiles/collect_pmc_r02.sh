# Round-2 PMC passes, one counter set per run (rocprofv3 --pmc only with --kernel-trace), on the bench command at a reduced
# batch (16384 reads per launch: a counter pass at the benched 100 000 reads takes ~7 minutes of box time; the per-band and
# per-read figures do not depend on the batch size -- profiles/r02_pmc_100k_pass1.json is the one pass that was run at 100 000).
# Output: gpurun_out/<tag>/pmc{1..4} -> profiles/pmc_summary.py
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; T=${1:-r02pmc}; O=$R/gpurun_out/$T; mkdir -p $O; cd /tmp
B="python $R/bench.py --steps 1 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 --pool 4096 --tile 4"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc1 -o p1 -- $B > $O/pmc1.log 2>&1; echo rc1=$?
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --output-format csv -d $O/pmc2 -o p2 -- $B > $O/pmc2.log 2>&1; echo rc2=$?
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc3 -o p3 -- $B > $O/pmc3.log 2>&1; echo rc3=$?
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc4 -o p4 -- $B > $O/pmc4.log 2>&1; echo rc4=$?
python3 $R/profiles/pmc_summary.py $O 16384 ${2:-13463.2} > $O/pmc.json; head -c 400 $O/pmc.json
