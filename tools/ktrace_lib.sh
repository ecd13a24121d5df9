#!/bin/bash
# tools/ktrace.sh with a given library build: tools/ktrace_lib.sh lib.so OUTNAME [bench args...]
lib=$1; shift
export DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_HIP_LIB=$PWD/$lib
exec tools/ktrace.sh "$@"
