cd $GRAFT_REPO_ROOT
OPTY_TUNE_NODES=12501 python tools/wave_timeline.py config3_10link fused auto > gpurun_out/r05_timeline_c3_shard.txt 2>&1
python tools/wave_timeline.py config3_10link fused auto > gpurun_out/r05_timeline_c3.txt 2>&1
python tools/wave_timeline.py config5_one_legged fused auto > gpurun_out/r05_timeline_leg_plan.txt 2>&1
grep -v amdgpu.ids gpurun_out/r05_timeline_c3_shard.txt gpurun_out/r05_timeline_c3.txt gpurun_out/r05_timeline_leg_plan.txt
timeout 900 python tools/tune_plans.py --tune config3_10link > gpurun_out/r05_tune3.txt 2>&1
grep "pad\|seed  \|world" gpurun_out/r05_tune3.txt
