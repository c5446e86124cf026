#!/bin/bash
# call 19: kernel B ablations -- emissions for free (upper bound of what a fused meth/unmeth pass could share), log-sums without the table
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
timeout 900 python tools/hmm_ab.py nanopolish_amd/variants/libnp_hip_hmm_base.so nanopolish_amd/variants/libnp_hip_hmm_noem.so nanopolish_amd/variants/libnp_hip_hmm_nolse.so > $O/hmm_ab.jsonl 2> $O/hmm_ab.err; cat $O/hmm_ab.jsonl; tail -3 $O/hmm_ab.err
