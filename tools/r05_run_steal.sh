cd $GRAFT_REPO_ROOT
python -m pytest tests/test_list_schedule.py -q -m gpu -x 2>&1 | tail -15
OPTY_AB_ROUNDS=5 python tools/ab_strips.py config5_one_legged auto "chunk=16,groups=5,fused_groups=5,order=tail,fused_strips=0:96;96:160;160:348,fused_order=list,park=48,park_live=235" auto+specialize "chunk=16,groups=5,fused_groups=5,order=tail,fused_strips=0:96;96:160;160:348,fused_order=list,park=48,park_live=235,specialize=1" 2>&1 | grep -v "emit_options give" | sed 's/chunk=16,groups=5,fused_groups=5,order=tail,fused_strips=0:96;96:160;160:348,//' | tail -6
