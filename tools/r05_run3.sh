cd $GRAFT_REPO_ROOT
export OPTY_AB_ROUNDS=5
python tools/ab_strips.py config5_one_legged auto "chunk=16,cut=work,groups=4,work_live=400,order=class" "chunk=16,cut=work,groups=4,work_live=400,order=class,park=48,park_live=235" "chunk=16,cut=work,groups=4,work_live=400,order=class,park=48,park_live=205" "chunk=16,cut=work,groups=4,work_live=400,order=class,park=48,park_live=215" "chunk=16,cut=work,groups=4,work_live=400,order=class,park=48,park_live=225" "chunk=16,cut=work,groups=4,work_live=400,order=block,park=48,park_live=235" > gpurun_out/r05_ab2_leg.txt 2>&1
python tools/wave_timeline.py config5_one_legged fused "chunk=16,cut=work,groups=4,work_live=400,order=class,park=48,park_live=235" "chunk=16,cut=work,groups=4,work_live=400,order=class,park=48,park_live=205" > gpurun_out/r05_timeline2_leg.txt 2>&1
grep -v "^emit_options" gpurun_out/r05_ab2_leg.txt gpurun_out/r05_timeline2_leg.txt
