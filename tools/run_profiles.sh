cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/p_* 
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p_stats -- python3 bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-fast-path > gpurun_out/p_stats_bench.json 2> gpurun_out/p_stats.err || exit 1
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p_fetch -- python3 tools/stage_times.py --poses 0 --reps 3 --tex-size 1024 > gpurun_out/p_fetch.log 2>&1 || exit 2
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p_write -- python3 tools/stage_times.py --poses 0 --reps 3 --tex-size 1024 > gpurun_out/p_write.log 2>&1 || exit 3
timeout -s KILL 180 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d gpurun_out/p_valu -- python3 tools/stage_times.py --poses 0 --reps 2 --tex-size 1024 > gpurun_out/p_valu.log 2>&1 || exit 4
timeout -s KILL 180 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 TA_TA_BUSY_sum --output-format csv -d gpurun_out/p_valu2 -- python3 tools/stage_times.py --poses 0 --reps 2 --tex-size 1024 > gpurun_out/p_valu2.log 2>&1 || exit 5
python3 tools/make_profiles.py --commit "${VKR_COMMIT:-}" --tag ${TAG:-r03} --out gpurun_out --fetch gpurun_out/p_fetch --write gpurun_out/p_write --valu gpurun_out/p_valu --valu2 gpurun_out/p_valu2 || exit 6
cp $(find gpurun_out/p_stats -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG:-r03}_bench4k_kernel_stats.csv
cp gpurun_out/p_stats_bench.json gpurun_out/${TAG:-r03}_bench4k_under_rocprof.json
rm -rf gpurun_out/p_stats gpurun_out/p_fetch gpurun_out/p_write gpurun_out/p_valu gpurun_out/p_valu2
echo PROFILES_OK
