cd $GRAFT_REPO_ROOT
cat /proc/loadavg
python -m pytest tests/test_hip_parity.py tests/test_varying_first_layout.py -q -m gpu -k "persistent or varying or host or pinned or fresh" 2>&1 | tail -3
echo "== quiet host: slices / fixed shares"
python tools/host_path_calls.py 40 2>&1 | tail -2
OPTY_HIP_SCATTER_SHARES=1 python tools/host_path_calls.py 40 2>&1 | tail -2
echo "== 96 busy loops on the host: slices / fixed shares / slices"
for i in $(seq 96); do (timeout 100 python -c "
while True: pass" &) ; done
sleep 2; cat /proc/loadavg
python tools/host_path_calls.py 40 2>&1 | tail -4
OPTY_HIP_SCATTER_SHARES=1 python tools/host_path_calls.py 40 2>&1 | tail -4
python tools/host_path_calls.py 40 2>&1 | tail -4
