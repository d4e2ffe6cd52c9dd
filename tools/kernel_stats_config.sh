# rocprofv3 kernel statistics of one bench configuration: bash tools/kernel_stats_config.sh <config> <label>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks_tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_tmp -- python3 bench.py --config $1 --steps 32 --warmup 8 --repeats 2 --no-cpu-baseline --no-fast-path > gpurun_out/ks_$2.json 2> gpurun_out/ks_$2.err || exit 1
cp $(find gpurun_out/ks_tmp -name "*kernel_stats.csv" | head -1) gpurun_out/ks_$2.csv
rm -rf gpurun_out/ks_tmp
echo "== $2"; head -12 gpurun_out/ks_$2.csv | cut -d, -f1-4,6,7 | cut -c1-110
