cd $GRAFT_REPO_ROOT
export OPTY_AB_ROUNDS=5
B="chunk=16,groups=5,order=tail,park=48,fused_strips=160:348+0:96+96:160"
S0="$B,park_live=235,park_spread=0"
S1="$B,park_live=245,park_spread=1"
S2="$B,park_live=250,park_spread=1"
if [ "$1" = "prebuild" ]; then
python tools/ab_strips.py config5_one_legged "$S0" "$S1" "$S2" 2>&1 | grep -v "^emit_options"
python tools/wave_timeline.py config5_one_legged fused "$S0" "$S1" 2>&1 | grep -v "^emit_options"
exit 0
fi
python tools/ab_strips.py config5_one_legged auto "$S0" "$S1" "$S2" 2>&1 | grep -v "^emit_options\|amdgpu.ids" > gpurun_out/r05_ab5.txt
python tools/wave_timeline.py config5_one_legged fused "$S0" "$S1" 2>&1 | grep -v "^emit_options\|amdgpu.ids" >> gpurun_out/r05_ab5.txt
cat gpurun_out/r05_ab5.txt
