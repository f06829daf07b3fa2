# developer tool: pose_lm over problem counts for experiment builds (tools/build_variant.sh <tag> opt_kernels.hip -DPO_T=128 ...; build_exp/libps_<tag>.so)
for V in "" po128 po64 poocc3 poocc4; do
  for N in 256 512 768 1024; do
    if [ -z "$V" ]; then L=""; else L="build_exp/libps_$V.so"; fi
    echo -n "variant=${V:-base} "; PS_LIB_PATH=$L python tools/pose_quick.py $N 2>/dev/null | tail -1
  done
done
