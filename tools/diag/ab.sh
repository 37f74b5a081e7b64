#!/bin/bash
# A/B two libraries on the same box, alternating.  A library whose name contains "_r1" is the round-1 ABI.
cd $GRAFT_REPO_ROOT
A=$1; B=$2; shift 2
leg() { case "$1" in *_r1*) echo 1;; *) echo 0;; esac; }
for i in 1 2 3; do
  echo "A:"; SFVOS_LEGACY=$(leg $A) SFVOS_LIB=$A timeout -k 10 120 python tools/diag/mb_conv.py "$@" || exit 1
  echo "B:"; SFVOS_LEGACY=$(leg $B) SFVOS_LIB=$B timeout -k 10 120 python tools/diag/mb_conv.py "$@" || exit 1
done
