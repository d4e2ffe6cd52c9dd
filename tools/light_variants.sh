# times the deferred lighting kernel of several builds of the library (exp/libvkr_v*.so) on the same poses
for v in "$@"; do
  export VKR_HIP_LIB=$PWD/exp/libvkr_v$v.so
  echo "variant $v"
  timeout -k 10 120 python tools/stage_times.py --poses 0,128,256,400 --reps 5 | sed -e 's/.*resolve_gbuf=\([0-9]*\).*deferred_light=\([0-9]*\).*/  gbuf \1 light \2/' || exit 1
done
