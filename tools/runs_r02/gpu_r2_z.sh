export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r02zo}; mkdir -p $O; cd $R
for cfg in "4 1" "8 2"; do set -- $cfg
echo "== kernel A $1 blocks/CU, kernel B $2 blocks/CU" >> $O/overlap.txt
NP_ALIGN_BLOCKS_PER_CU=$1 NP_HMM_BLOCKS_PER_CU=$2 timeout 100 python tools/overlap_ab.py --parts 2 --steps 4 2>&1 | grep "\"one\"\|stages\|equal\|Error\|error" >> $O/overlap.txt
done
cut -c1-170 $O/overlap.txt
