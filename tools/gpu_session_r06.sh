python -m pytest tests/test_routing.py tests/test_shard_host.py tests/test_varying_first_layout.py -m gpu -q -x > gpurun_out/gputest3.log 2>&1; tail -5 gpurun_out/gputest3.log
python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "host or persistent or varying or known_maps or refused_persistent" >> gpurun_out/gputest3.log 2>&1; tail -3 gpurun_out/gputest3.log
for v in "" "OPTY_HIP_COPY_STREAMS=1" "OPTY_HIP_HOST_TAPER=0" "OPTY_HIP_COPY_STREAMS=1 OPTY_HIP_HOST_TAPER=0"; do echo "== $v" >> gpurun_out/host_trace3.txt; env $v python tools/host_path_trace.py 2>&1 >/dev/null | grep -E "^call 1[0-9]|chunks:" | tail -8 >> gpurun_out/host_trace3.txt; done
tail -30 gpurun_out/host_trace3.txt
for w in config5_one_legged config5_biped; do python tools/ab_strips.py $w auto auto+share_rcp auto+specialize auto+specialize+share_rcp >> gpurun_out/ab_share_rcp.txt 2>&1; done
OPTY_TUNE_NODES=6251 python tools/ab_strips.py config5_one_legged auto auto+share_rcp >> gpurun_out/ab_share_rcp.txt 2>&1
cat gpurun_out/ab_share_rcp.txt | grep -v "^$" | tail -40
python tools/agpr_experiment.py > gpurun_out/agpr_experiment.txt 2>&1; tail -25 gpurun_out/agpr_experiment.txt
