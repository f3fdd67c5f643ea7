cd $GRAFT_REPO_ROOT
export OPTY_AB_ROUNDS=5
python tools/ab_strips.py config5_one_legged auto "chunk=16,groups=5,fused_groups=5,order=class" "chunk=16,groups=5,fused_groups=5,order=class,ablate=uni_lit" "chunk=16,cut=work,groups=4,work_live=400,order=class" "chunk=16,cut=work,groups=4,work_live=400,order=block" "chunk=32,cut=work,groups=4,work_live=400,order=class" "chunk=16,cut=work,groups=5,order=class" > gpurun_out/r05_ab1_leg.txt 2>&1
python tools/ab_strips.py config5_biped "cut=work,groups=5,fused_groups=5,order=block" "cut=work,groups=5,fused_groups=5,order=class,ablate=uni_lit" "cut=work,groups=4,fused_groups=4,work_live=330,order=class" "cut=work,groups=3,fused_groups=3,work_live=330,order=class" "cut=work,groups=2,fused_groups=2,work_live=450,order=class" "cut=work,groups=4,fused_groups=4,work_live=450,order=class" auto > gpurun_out/r05_ab1_biped.txt 2>&1
tail -12 gpurun_out/r05_ab1_leg.txt gpurun_out/r05_ab1_biped.txt
