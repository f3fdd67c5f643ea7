cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_deterministic.py tests/test_c_client.py "tests/test_hip_parity.py::test_row_sorted_muscle_model_o1_twin_is_the_faulty_one" "tests/test_hip_parity.py::test_biped_build_with_twenty_strips_is_refused" "tests/test_hip_parity.py::test_spilling_parked_wave_is_refused" tests/test_hip_parity.py -k "deterministic or c_client or faulty or refused or specialised or fresh_to_the_caller or known_maps" -x -q -m gpu > gpurun_out/r05_newtests.txt 2>&1
tail -15 gpurun_out/r05_newtests.txt
timeout 1500 python tools/tune_plans.py --tune config5_one_legged config5_biped > gpurun_out/r05_tune1.txt 2>&1
tail -5 gpurun_out/r05_tune1.txt
OPTY_SOAK_DETERMINISTIC=1 timeout 600 python tools/window_soak.py 120 > gpurun_out/r05_det_soak.txt 2>&1
tail -3 gpurun_out/r05_det_soak.txt
