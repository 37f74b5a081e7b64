cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv3d" 2>&1 | tail -2
for i in 1 2 3; do
 for lib in lib_base lib_fs1; do
  echo "== $lib"
  SFVOS_LIB=scratch/$lib.so timeout -k 10 120 python tools/diag/mb_conv.py f3 50 2>&1 | grep "conv " || exit 1
 done
done
