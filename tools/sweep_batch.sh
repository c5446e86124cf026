#!/bin/bash
# batch-size sweep: reads per step = 1024 * tile; 7168 waves are resident (7 blocks x 4 waves x 256 CUs)
for t in ${TILES:-7 14 16 21 28 32}; do
  echo -n "tile=$t "
  timeout 250 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --tile $t 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); k=d["roofline"]["kernel_ms_per_step"]; n=d["config"]["reads_per_step_per_gpu"]; print(n, d["value"], d["ms_per_step"], k, "align us/read %.3f hmm us/read %.3f" % (k["event_align"]*1e3/n, k["hmm_score"]*1e3/n))'
done
