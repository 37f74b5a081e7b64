cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
 for v in "" 1; do
  echo "== NT2=$v"
  env ${v:+SFVOS_DG_NT2=1} SFVOS_LIB=scratch/lib_dgnt2.so timeout -k 10 120 python tools/diag/mb_conv.py wide 20 2>&1 | grep "dgrad" || exit 1
 done
done
