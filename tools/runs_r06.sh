#!/bin/bash
# Round-6 GPU calls, one function per call (provenance of the gpurun tags the files under profiles/ cite: r06a ...).
#   usage on the GPU box (through gpurun):  bash tools/runs_r06.sh <letter>        e.g.  gpurun -- 'bash tools/runs_r06.sh b'
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; export GRAFT_REPO_ROOT=$R
V=nanopolish_amd/variants

# r06a was: ./tools/mfma_f64_probe (the matrix pipe as kernel A's fp64 adder: layout, exactness, rates)

# the GPU suite after the C-ABI changes (explicit ADC verdicts, host scoring vs a declared layout), the probe with the scalar-store trace
# mode, v_readlane's lane-select forms, and kernel A with one band-end lane select in M0 against the shipped build (A/B/A/B, same box)
call_b() {
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
./tools/mfma_f64_probe > $O/probe.log 2>&1
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/valu_rates.hip -o /tmp/valu_rates > /dev/null 2>&1 && /tmp/valu_rates l > $O/valu_l.log 2>&1
( timeout 900 python tools/align_ab.py --pool 2048 --tile 16 $V/libnp_hip_base.so $V/libnp_hip_m0sel.so $V/libnp_hip_base.so $V/libnp_hip_m0sel.so ) > $O/align_ab.log 2>&1
( time timeout 600 python bench.py --steps 5 --warmup 2 --legs 0 --streamed 0 --ragged 0 --cpu-sample 0 ) > $O/bench.json 2> $O/bench.err
tail -5 $O/pytest.log; grep "waves/SIMD 8" $O/probe.log; cat $O/valu_l.log; cat $O/align_ab.log; head -c 600 $O/bench.json; tail -3 $O/bench.err
}

"call_$1"
