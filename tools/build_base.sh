#!/bin/bash
# Builds the library of a COMMIT (default HEAD) to build_tune/lib_<name>.so for a same-box A/B against the working tree
# (tools/ab_round.sh "WG_LIB=$PWD/build_tune/lib_<name>.so" ""):   bash tools/build_base.sh [commit] [name]
set -e
C=${1:-HEAD}; NAME=${2:-base}; R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
git -C $R archive $C wittgenstein_amd/csrc include | tar -x -C $T
mkdir -p $R/build_tune
(cd $T/wittgenstein_amd/csrc && bash build.sh -o $R/build_tune/lib_$NAME.so 2>&1 | grep -E "error" || true)
rm -rf $T; ls -la $R/build_tune/lib_$NAME.so
