cd $GRAFT_REPO_ROOT
export OPTY_AB_ROUNDS=5
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_c_abi_errors.py -k "plan_flags or tune_launch or fresh_to_the_caller" -q -m gpu 2>&1 | tail -3
for w in config3_10link config5_one_legged config5_biped config5_standin_24link; do
  echo "== $w" >> gpurun_out/r05_ab_deterministic.txt
  python tools/ab_strips.py $w auto auto+deterministic 2>&1 | grep -v "^emit_options\|amdgpu.ids\|^kernels" >> gpurun_out/r05_ab_deterministic.txt
done
cat gpurun_out/r05_ab_deterministic.txt
timeout 2400 python tools/tune_plans.py --tune config3_10link config5_standin_24link config5_gaitlike_24link config5_one_legged config5_biped config2_pendulum > gpurun_out/r05_tune4.txt 2>&1
grep -v "^    " gpurun_out/r05_tune4.txt | grep -v amdgpu | tail -14
OPTY_WORKLOAD=config5_one_legged bash tools/pmc.sh default r05_pmc_leg fused > gpurun_out/r05_pmc_leg.txt 2>&1
OPTY_WORKLOAD=config5_one_legged OPTY_SPECIALIZE=1 bash tools/pmc.sh default r05_pmc_leg_spec fused > gpurun_out/r05_pmc_leg_spec.txt 2>&1
OPTY_WORKLOAD=config5_biped bash tools/pmc.sh default r05_pmc_biped fused > gpurun_out/r05_pmc_biped.txt 2>&1
tail -30 gpurun_out/r05_pmc_leg.txt
