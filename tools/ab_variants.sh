# A/B of library builds (exp/libvkr_v<name>.so; "default" = the in-tree library): per-stage times over four poses at the
# benchmark's texture size, then one FETCH_SIZE counter pass each (L2 misses per kernel, pose 0).
#   bash tools/ab_variants.sh default noswz
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = default ]; then unset VKR_HIP_LIB; else export VKR_HIP_LIB=$PWD/exp/libvkr_v$v.so; fi
  echo "== variant $v: stage times (us)"
  timeout -k 10 200 python3 tools/stage_times.py --poses 0,128,256,400 --reps 5 --tex-size 1024 || exit 1
done
for v in "$@"; do
  if [ "$v" = default ]; then unset VKR_HIP_LIB; else export VKR_HIP_LIB=$PWD/exp/libvkr_v$v.so; fi
  rm -rf gpurun_out/ab_fetch_$v
  timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/ab_fetch_$v -- python3 tools/stage_times.py --poses 0 --reps 3 --tex-size 1024 > /dev/null 2>&1 || exit 2
  echo "== variant $v: FETCH_SIZE per launch, doubled (MB)"
  python3 - gpurun_out/ab_fetch_$v <<'PY'
import sys
sys.path.insert(0, "tools")
from make_profiles import per_kernel, avg
for k, c in sorted(per_kernel(sys.argv[1]).items()):
    if "FETCH_SIZE" in c and len(c["FETCH_SIZE"]) > 1:
        print("  %-28s %8.1f" % (k, 2 * avg(c["FETCH_SIZE"]) * 1024 / 1e6))
PY
  rm -rf gpurun_out/ab_fetch_$v
done
