#!/bin/bash
# kernel-tuning variant of the library: tools/build_variant.sh <name> [-DFLAG ...]  ->  microwakeword_amd/libmww_<name>.so
# (default-topology kernels only, -DMWW_SLIM; FULL=1 in the environment builds every shape, e.g. for --model inception / notebook;
# add -DMWW_PROFILE for the ablation / phase-clock switches).  Same translation units and object cache as the product build.
R=$(cd $(dirname $0)/.. && pwd)
N=$1; shift
SLIM=--slim; [ -n "$FULL" ] && SLIM=
cd $R && python -m microwakeword_amd.build_native --out $R/microwakeword_amd/libmww_$N.so $SLIM -- "$@" 2>&1 | grep -E "error|Error|compiled" ; ls -la $R/microwakeword_amd/libmww_$N.so
