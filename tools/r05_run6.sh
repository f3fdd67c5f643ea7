cd $GRAFT_REPO_ROOT
export OPTY_AB_ROUNDS=5
timeout 1800 python -m pytest tests/test_deterministic.py tests/test_c_client.py tests/test_hip_parity.py -k "deterministic or same_bits or c_client or faulty or refused or specialised or fresh_to_the_caller or known_maps or golden_full" -q -m gpu > gpurun_out/r05_newtests.txt 2>&1
tail -12 gpurun_out/r05_newtests.txt
for w in config5_one_legged config5_biped config5_standin_24link; do
  python tools/ab_strips.py $w auto auto+specialize 2>&1 | grep -v "^emit_options\|amdgpu.ids" >> gpurun_out/r05_ab_specialize.txt
done
cat gpurun_out/r05_ab_specialize.txt
timeout 2400 python tools/tune_plans.py --tune config5_one_legged config5_biped config3_10link config5_standin_24link config5_gaitlike_24link > gpurun_out/r05_tune2.txt 2>&1
grep -v "^    " gpurun_out/r05_tune2.txt | tail -14
