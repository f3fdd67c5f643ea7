cd $GRAFT_REPO_ROOT
for w in config5_one_legged config3_10link config5_biped config5_standin_24link; do
  echo "== $w"; OPTY_AB_ROUNDS=5 python tools/ab_strips.py $w auto auto+uniform_trig 2>&1 | grep -v "emit_options give\|amdgpu" | tail -2
done
