#!/bin/bash
# occupancy sweep of kernel A: blocks (4 waves each) resident per CU
for n in 3 4 5 6 7 8; do
  echo -n "blocks_per_cu=$n "
  NP_ALIGN_BLOCKS_PER_CU=$n timeout 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["roofline"]["kernel_ms_per_step"])'
done
