#!/bin/bash
# A/B two libraries on the same box, alternating
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
for i in 1 2 3; do
  echo "A:"; SFVOS_LIB=$A timeout -k 10 120 python tools/diag/mb_conv.py "$@" || exit 1
  echo "B:"; SFVOS_LIB=$B timeout -k 10 120 python tools/diag/mb_conv.py "$@" || exit 1
done
