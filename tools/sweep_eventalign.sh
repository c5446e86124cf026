# occupancy / batch sweep of the eventalign chain kernel (run through gpurun): "<waves per CU> <tile>" pairs
mkdir -p gpurun_out
for cfg in "16 32" "20 40" "20 80" "16 64"; do set -- $cfg
NP_EA_WAVES_PER_CU=$1 timeout 600 python tests/bench_eventalign.py --pool 256 --tile $2 --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('waves/CU $1 tile $2:', d['value'], 'reads/s', d['ms_per_step'], 'ms/step', d['kernel_ms_per_step'])"
done
