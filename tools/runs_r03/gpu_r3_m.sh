#!/bin/bash
# call 17: kernel timeline of the pipelined pass (which kernels of the two streams run side by side?)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O; cd /tmp
for bt in 4; do
  timeout 600 rocprofv3 --kernel-trace -d $O/tr$bt -o t -- python $R/tools/split_align_bench.py --steps 3 --modes pipelined --bt-blocks $bt > $O/run$bt.log 2>&1
  tail -2 $O/run$bt.log | cut -c1-400
  f=$(find $O/tr$bt -name "*results.db" | head -1); [ -n "$f" ] && python3 $R/profiles/timeline_rocpd.py $f --min-ms 2 --last 40 > $O/timeline$bt.md
  rm -rf $O/tr$bt
  cat $O/timeline$bt.md | cut -c1-260
done
