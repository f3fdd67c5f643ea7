cd $GRAFT_REPO_ROOT
export OPTY_AB_ROUNDS=5
L0="chunk=16,cut=work,groups=4,work_live=400,order=block,park=48,park_live=235"
L1="chunk=16,strips=96:160;160:348+0:96,order=class,park=48,park_live=235"
L2="chunk=16,strips=96:160;160:348+0:96,order=block,park=48,park_live=235"
L3="chunk=16,cut=work,groups=4,work_live=400,order=tail,park=48,park_live=235"
L4="chunk=16,cut=work,groups=4,work_live=400,order=block,park=48,park_live=235,flush_unroll=8"
L5="chunk=16,strips=96:160;160:256;256:348+0:96,order=class,park=48,park_live=235"
L6="chunk=16,cut=work,groups=5,fused_groups=4,work_live=400,order=class,park=48,park_live=235"
B0="cut=work,groups=5,fused_groups=5,order=block"
B1="cut=work,groups=5,fused_groups=5,order=tail"
B2="cut=work,groups=5,fused_groups=5,order=block,flush_unroll=8"
B3="chunk=16,cut=work,groups=5,fused_groups=5,order=tail"
if [ "$1" = "prebuild" ]; then
python tools/ab_strips.py config5_one_legged "$L0" "$L1" "$L2" "$L3" "$L4" "$L5" "$L6" 2>&1 | grep -v "^emit_options"
python tools/ab_strips.py config5_biped "$B0" "$B1" "$B2" "$B3" 2>&1 | grep -v "^emit_options"
python tools/wave_timeline.py config5_one_legged fused "$L1" 2>&1 | grep -v "^emit_options"
python tools/wave_timeline.py config5_biped fused "$B1" 2>&1 | grep -v "^emit_options"
exit 0
fi
python tools/ab_strips.py config5_one_legged auto "$L0" "$L1" "$L2" "$L3" "$L4" "$L5" "$L6" > gpurun_out/r05_ab3_leg.txt 2>&1
python tools/ab_strips.py config5_biped "$B0" "$B1" "$B2" "$B3" > gpurun_out/r05_ab3_biped.txt 2>&1
python tools/wave_timeline.py config5_one_legged fused "$L1" > gpurun_out/r05_timeline3_leg.txt 2>&1
python tools/wave_timeline.py config5_biped fused "$B1" > gpurun_out/r05_timeline3_biped.txt 2>&1
grep -v "^emit_options" gpurun_out/r05_ab3_leg.txt gpurun_out/r05_ab3_biped.txt gpurun_out/r05_timeline3_leg.txt gpurun_out/r05_timeline3_biped.txt
