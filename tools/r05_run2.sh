cd $GRAFT_REPO_ROOT
python tools/wave_timeline.py config5_one_legged fused "chunk=16,groups=5,fused_groups=5" "chunk=16,groups=5,fused_groups=5,order=class" "chunk=16,cut=work,groups=4,work_live=400,order=class" "chunk=32,cut=work,groups=4,work_live=400,order=class" "chunk=16,groups=5,fused_groups=5,ablate=store_only" > gpurun_out/r05_timeline_leg.txt 2>&1
python tools/wave_timeline.py config5_biped fused "cut=work,groups=5,fused_groups=5,order=block" "cut=work,groups=5,fused_groups=5,order=class" > gpurun_out/r05_timeline_biped.txt 2>&1
cat gpurun_out/r05_timeline_leg.txt gpurun_out/r05_timeline_biped.txt | grep -v "^emit_options" | tail -120
