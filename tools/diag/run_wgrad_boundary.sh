# timing-only: how much of the frame-paired weight gradient is the ring refill at tile boundaries (-DSFVOS_WG_ABLATE=8: libsfvos_wabl8.so) and all staging copies (=12)?  Result (r3): 2.284 -> 2.246 ms (-1.7 %) / 2.045 ms (-10.5 %)
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/applying-slowfast-networks-to-video-object-segmentation_amd/csrc
for i in 1 2 3; do
  echo shipped; timeout -k 10 60 python tools/diag/mb_conv.py wf1 30 2>&1 | grep "^wgrad"
  echo no_boundary_refill; SFVOS_LIB=$L/libsfvos_wabl8.so timeout -k 10 60 python tools/diag/mb_conv.py wf1 30 2>&1 | grep "^wgrad"
  echo no_copies_at_all; SFVOS_LIB=$L/libsfvos_wabl12.so timeout -k 10 60 python tools/diag/mb_conv.py wf1 30 2>&1 | grep "^wgrad"
done
