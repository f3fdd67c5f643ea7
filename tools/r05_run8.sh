cd $GRAFT_REPO_ROOT
export OPTY_AB_ROUNDS=5
L="chunk=16,groups=5,order=tail,park=48,park_live=235"
LA="$L,fused_strips=0:96;96:160;160:256;256:348"
LB="$L,fused_strips=0:96;96:160;160:224;224:288;288:348"
LC="$L,fused_strips=0:96;96:160;160:256;256:348,fused_order=tail"
LD="$L,fused_strips=0:96;96:160;160:348,fused_order=tail"
B="chunk=16,cut=work,groups=5,order=tail,fused_order=tail"
BA="$B,fused_strips=0:224;224:448;448:528;528:624;624:774,park=48,park_live=238"
BB="$B,fused_strips=0:224;224:448;448:528;528:624;624:774,park=48,park_live=230"
if [ "$1" = "prebuild" ]; then
python tools/ab_strips.py config5_one_legged auto "$LA" "$LB" "$LC" "$LD" 2>&1 | grep -v "^emit_options"
python tools/ab_strips.py config5_biped auto "$BA" "$BB" 2>&1 | grep -v "^emit_options"
exit 0
fi
python tools/ab_strips.py config5_one_legged auto "$LA" "$LB" "$LC" "$LD" 2>&1 | grep -v "^emit_options\|amdgpu.ids" > gpurun_out/r05_ab4.txt
python tools/ab_strips.py config5_biped auto "$BA" "$BB" 2>&1 | grep -v "^emit_options\|amdgpu.ids" >> gpurun_out/r05_ab4.txt
cat gpurun_out/r05_ab4.txt
