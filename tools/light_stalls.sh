# counter passes that attribute the lighting kernel's cycles (separate runs; SQ block holds 8 counters per run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ls_a gpurun_out/ls_b
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/ls_a -- python3 tools/stage_times.py --poses 0 --reps 2 > gpurun_out/ls_a.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_INST_LEVEL_SMEM SQ_IFETCH GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/ls_b -- python3 tools/stage_times.py --poses 0 --reps 2 > gpurun_out/ls_b.log 2>&1 || exit 2
python3 tools/pmc_summary.py gpurun_out/ls_a gpurun_out/ls_b > gpurun_out/${TAG:-r02}_light_stall_counters.txt
rm -rf gpurun_out/ls_a gpurun_out/ls_b
