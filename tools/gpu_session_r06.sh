#!/bin/bash
# scratch driver of one GPU lease (r06): pieces selected by arguments, every
# piece under its own timeout, output under gpurun_out/
mkdir -p gpurun_out
for piece in "$@"; do
case $piece in
tests_new)
  timeout 900 python -m pytest tests/test_routing.py tests/test_shard_host.py tests/test_isomorph.py tests/test_varying_first_layout.py -m gpu -q --timeout 240 > gpurun_out/gputest3.log 2>&1; tail -15 gpurun_out/gputest3.log;;
tests_host)
  timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout 300 -k "host or persistent or varying or known_maps or refused_persistent or wrong_high" > gpurun_out/gputest3b.log 2>&1; tail -5 gpurun_out/gputest3b.log;;
trace)
  rm -f gpurun_out/host_trace3.txt
  for v in "X=1" "OPTY_HIP_COPY_STREAMS=1" "OPTY_HIP_HOST_TAPER=0" "OPTY_HIP_COPY_STREAMS=1 OPTY_HIP_HOST_TAPER=0"; do echo "== $v" >> gpurun_out/host_trace3.txt; env $v timeout 300 python tools/host_path_trace.py 2>&1 >/dev/null | grep -E "^call 1[0-9]|chunks:|jacobian call" | tail -8 >> gpurun_out/host_trace3.txt; done
  cat gpurun_out/host_trace3.txt | cut -c1-200;;
ab_rcp)
  rm -f gpurun_out/ab_share_rcp.txt
  for w in config5_one_legged config5_biped; do timeout 600 python tools/ab_strips.py $w auto auto+share_rcp auto+specialize auto+specialize+share_rcp >> gpurun_out/ab_share_rcp.txt 2>&1; done
  OPTY_TUNE_NODES=6251 timeout 600 python tools/ab_strips.py config5_one_legged auto auto+share_rcp >> gpurun_out/ab_share_rcp.txt 2>&1
  grep -v "^$" gpurun_out/ab_share_rcp.txt | tail -30;;
ab_publish)
  timeout 900 python tools/ab_publish.py > gpurun_out/ab_publish.txt 2>&1; tail -12 gpurun_out/ab_publish.txt;;
agpr)
  timeout 1200 python tools/agpr_experiment.py > gpurun_out/agpr_experiment.txt 2>&1; tail -25 gpurun_out/agpr_experiment.txt | cut -c1-250;;
rehearsal)
  timeout 1500 tools/scale_rehearsal.sh gpurun_out/rehearsal > gpurun_out/rehearsal.log 2>&1; tail -40 gpurun_out/rehearsal.log;;
esac
done
