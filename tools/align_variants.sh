#!/bin/bash
# builds the A/B variants of kernel A (run in the build container; the .so files travel with gpurun)
cd "$(dirname "$0")/../nanopolish_amd/csrc" || exit 1
make -s -j8 || exit 1
rm -f ../variants/*
v() { make -s variant TAG=$1 UNIT=np_align_kernel VARFLAGS="$2" 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^\|generated" ; }
v all      ""
v u8off    "-DNP_A_UNROLL8=0"
v cmpxoff  "-DNP_A_CMPX=0"
v traceoff "-DNP_A_TRACEASM=0"
v nobt     "-DNP_ABL=128"
ls ../variants/*.so | wc -l
