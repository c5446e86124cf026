#!/bin/bash
# round-2 first GPU pass: parity tests, the literal bench line, the 2-rank rehearsal on one GPU, a kernel-trace profile
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02a; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
( time timeout 900 python bench.py --steps 3 --warmup 1 ) > $O/bench.log 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
( time NP_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --pool 2000 --tile 5 --cpu-sample 0 --streamed 0 --ragged 0 ) > $O/bench_2rank_gloo.log 2> $O/bench_2rank_gloo.err
( time timeout 600 python bench.py --gpus 1 --pool 4000 --tile 5 --cpu-sample 0 --streamed 0 --ragged 0 ) > $O/bench_1rank_4000.log 2> $O/bench_1rank_4000.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample 0 --streamed 0 --ragged 0 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R $O/prof | head -30 >> $O/prof.log
tail -5 $O/pytest.log; tail -c 3000 $O/bench.log; tail -5 $O/bench.err
f=$(find $O/prof -name "*results.db" | head -1); [ -n "$f" ] && python3 profiles/summarize_rocpd.py $f > $O/prof_summary.md; head -20 $O/prof_summary.md
