#!/bin/bash
# round-2 second GPU pass: full parity suite, kernel A variants + ablations, ragged issue order, the bench line, host facts
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; lscpu | grep -i "model name\|socket\|thread\|^CPU(s)\|NUMA node(s)"; cat /proc/loadavg; free -g | head -2 ) > $O/host.txt 2>&1
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
V=nanopolish_amd/variants
timeout 900 python tools/align_ab.py $V/libnp_hip_base.so $V/libnp_hip_intc.so $V/libnp_hip_onerec.so $V/libnp_hip_ddbl.so $V/libnp_hip_all.so \
    $V/libnp_hip_abl1.so $V/libnp_hip_abl2.so $V/libnp_hip_abl3.so $V/libnp_hip_abl4.so $V/libnp_hip_abl5.so > $O/align_ab.txt 2>&1
( NP_ALIGN_LPT=0 timeout 300 python tools/align_ab.py --ragged 1 $V/libnp_hip_all.so; NP_ALIGN_LPT=1 timeout 300 python tools/align_ab.py --ragged 1 $V/libnp_hip_all.so ) > $O/align_ragged.txt 2>&1
( time timeout 900 python bench.py --steps 3 --warmup 1 ) > $O/bench.log 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_streamed -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --pool 4000 --tile 5 --cpu-sample 0 --ragged 0 > $GRAFT_REPO_ROOT/$O/prof_streamed.log 2>&1
cd $GRAFT_REPO_ROOT
tail -15 $O/pytest.log; cat $O/host.txt; cat $O/align_ab.txt $O/align_ragged.txt; tail -c 2500 $O/bench.log; tail -3 $O/bench.err
