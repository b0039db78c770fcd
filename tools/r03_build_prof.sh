#!/bin/bash
# the two instrumented builds tools/r03_large.sh reads (phase split, Riccati sub-phase split); objects are removed first: make does
# not see a change of -D flags
cd "$(dirname "$0")/../iterativelqr.jl_amd/csrc" || exit 1
rm -f ../lib_sub/*.o ../lib_sub/*.so ../lib_phase/*.o ../lib_phase/*.so
make -s LIBDIR=../lib_sub EXTRA="-DILQR_PROFILE -DILQR_PROFILE_SUB" &
make -s LIBDIR=../lib_phase EXTRA="-DILQR_PROFILE"
wait
ls -la ../lib_sub/libilqr_hip.so ../lib_phase/libilqr_hip.so
