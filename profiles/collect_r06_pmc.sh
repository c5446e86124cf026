#!/bin/bash
# Round-6 counter passes over the SHIPPED kernels A (np_event_align_kernel), B (np_hmm_forward_kernel: the call-methylation mix and,
# in passes of its own, the variants shape) and the eventalign chain (np_eventalign_chain2_kernel), at a launch size a counter pass
# finishes at.  Run on the GPU box through gpurun:   bash profiles/collect_r06_pmc.sh [tag] [reads per launch]
# One process per pass (rocprofv3 --pmc only together with --kernel-trace; FETCH_SIZE and WRITE_SIZE cannot share a pass).
# Summary: profiles/pmc_summary_r06.py (run in the build container: it prices the kernels' instruction mix from the assembly).
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${1:-r06pmc}; N=${2:-8192}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
W="python $R/tools/pmc_workload.py --reads $N --ea-reads $N --reps 2"
V="python $R/tools/pmc_workload.py --var-tile 1 --reps 2"
( cd /tmp && timeout 200 $W --timing-reps 5 > $O/units.json 2> $O/units.err ); echo "units rc=$?"        # the run WITHOUT counters: a warm-up step, five timed ones
( cd /tmp && timeout 200 $V --timing-reps 5 > $O/units_variants.json 2> $O/units_variants.err ); echo "units variants rc=$?"
pass() {   # name, workload, counters...
  local name=$1; local wl=$2; shift; shift
  ( cd /tmp && timeout ${PASS_TIMEOUT:-200} rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$name -o $name -- $wl > $O/$name.log 2>&1 ); echo "$name rc=$?" | tee -a $O/passes.log
}
pass sq1 "$W" SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass fetch "$W" FETCH_SIZE
pass write "$W" WRITE_SIZE
pass sq2 "$W" SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass var_sq1 "$V" SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
pass var_fetch "$V" FETCH_SIZE
pass var_write "$V" WRITE_SIZE
cat $O/passes.log; tail -1 $O/units.json | cut -c1-600; tail -1 $O/units_variants.json
