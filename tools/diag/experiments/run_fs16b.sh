cd $GRAFT_REPO_ROOT
export SFVOS_LIB=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc/libsfvos_diag.so
for i in 1 2 3; do
  echo "== 16-px, 8+8+4+2"; timeout -k 10 120 python tools/diag/mb_conv.py f1 20 2>&1 | grep "^conv"
  echo "== 16-px, 8+8+8"; SFVOS_FS16_PAD=1 timeout -k 10 120 python tools/diag/mb_conv.py f1 20 2>&1 | grep "^conv"
  echo "== 32-px"; SFVOS_FS_TW32=1 timeout -k 10 120 python tools/diag/mb_conv.py f1 20 2>&1 | grep "^conv"
done
for v in "" SFVOS_FS16_PAD SFVOS_FS_TW32; do
  echo "== bench $v"; env ${v:+$v=1} timeout -k 10 200 python bench.py --no-cpu-baseline --no-dropin --steps 20 --warmup 4 > gpurun_out/fs16_b.json 2>/dev/null; python tools/diag/show_bench.py gpurun_out/fs16_b.json 2 | grep -v roofline
done
