cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
 for lib in lib_pda3 lib_pda5 lib_pda8; do
  echo "== $lib"
  SFVOS_LIB=scratch/$lib.so timeout -k 10 120 python tools/diag/mb_conv.py f1 10 2>&1 | grep "conv " || exit 1
  SFVOS_LIB=scratch/$lib.so timeout -k 10 120 python tools/diag/mb_conv.py f2 30 2>&1 | grep "conv " || exit 1
 done
done
