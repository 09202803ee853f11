#!/bin/bash
# usage: scripts/build_ab.sh <name> [-DFLAG ...]   ->  nestfit_amd/lib/ab_<name>.so  (an A/B build of the engine:
# the same sources with experiment macros; selected at run time with NFA_ENGINE_LIB)
name=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
    "$@" -o nestfit_amd/lib/ab_$name.so nestfit_amd/csrc/nfa_engine.hip
