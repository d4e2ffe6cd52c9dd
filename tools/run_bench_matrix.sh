# Bench lines of the round (one GPU): the default line, the five BASELINE.json configurations, two-rank rehearsals (gloo,
# ranks sharing the GPU) of both partitions.  Usage: bash tools/run_bench_matrix.sh <tag>; output under gpurun_out/.
TAG=${1:-r02}
O=gpurun_out
set -e
timeout -k 10 200 python bench.py > $O/${TAG}_bench4k_default_run.json 2> $O/${TAG}_default.err
for c in 1 2 3 4 5; do
  timeout -k 10 200 python bench.py --config $c > $O/${TAG}_config${c}_run.json 2> $O/${TAG}_config${c}.err
done
timeout -k 10 200 python bench.py --fused --no-cpu-baseline > $O/${TAG}_bench4k_fused_run.json 2> $O/${TAG}_fused.err
timeout -k 10 200 python bench.py --lanes 2 --no-cpu-baseline > $O/${TAG}_bench4k_lanes2_run.json 2> $O/${TAG}_lanes2.err
timeout -k 10 200 python bench.py --fused --lanes 2 --no-cpu-baseline > $O/${TAG}_bench4k_lanes2_fused_run.json 2> $O/${TAG}_lanes2_fused.err
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --no-cpu-baseline > $O/${TAG}_2rank_gloo_poses.json 2> $O/${TAG}_2rank_poses.err
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --partition sortfirst --no-cpu-baseline > $O/${TAG}_2rank_gloo_sortfirst_bands.json 2> $O/${TAG}_2rank_sf.err
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --partition sortfirst --stripe-height 64 --no-cpu-baseline > $O/${TAG}_2rank_gloo_sortfirst_stripes.json 2> $O/${TAG}_2rank_sfs.err
echo MATRIX_OK
