cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv3d" 2>&1 | tail -3
for i in 1 2 3; do
 for lib in lib_base lib_warm; do
  echo "== $lib"
  SFVOS_LIB=scratch/$lib.so timeout -k 10 120 python tools/diag/mb_conv.py f1 10 2>&1 | grep "conv " || exit 1
 done
done
