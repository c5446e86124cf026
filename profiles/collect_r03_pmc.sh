#!/bin/bash
# Round-3 counter passes over the SHIPPED kernels A (np_event_align_kernel), B (np_hmm_forward_kernel) and the eventalign chain
# (np_eventalign_chain_kernel) at a launch size a counter pass finishes at.  Run on the GPU box through gpurun:
#     bash profiles/collect_r03_pmc.sh [tag] [reads per launch]
# One process per pass (rocprofv3 --pmc only together with --kernel-trace; FETCH_SIZE and WRITE_SIZE cannot share a pass), the
# workload is tools/pmc_workload.py.  Summary: profiles/pmc_summary_r03.py -> profiles/r03_pmc.json.
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${1:-r03pmc}; N=${2:-2048}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
W="python $R/tools/pmc_workload.py --reads $N --ea-reads $N --reps 2"
( cd /tmp && timeout 120 $W > $O/units.json 2> $O/units.err ); echo "units rc=$?"
pass() {   # name, counters...
  local name=$1; shift
  ( cd /tmp && timeout ${PASS_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o $name -- $W > $O/$name.log 2>&1 ); echo "$name rc=$?" | tee -a $O/passes.log
}
pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
python3 profiles/pmc_summary_r03.py $O > $O/r03_pmc.json 2> $O/summary.err
head -c 1500 $O/r03_pmc.json; tail -2 $O/summary.err
