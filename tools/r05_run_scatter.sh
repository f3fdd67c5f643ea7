cd $GRAFT_REPO_ROOT
cat /proc/loadavg
for P in 0 1; do
  echo "== prune_zeros=$P slices"; HOST_CALLS_PRUNE=$P python tools/host_path_calls.py 40 2>&1 | tail -2 | head -1
  echo "== prune_zeros=$P fixed shares"; HOST_CALLS_PRUNE=$P OPTY_HIP_SCATTER_SHARES=1 python tools/host_path_calls.py 40 2>&1 | tail -2 | head -1
  echo "== prune_zeros=$P slices"; HOST_CALLS_PRUNE=$P python tools/host_path_calls.py 40 2>&1 | tail -2 | head -1
  echo "== prune_zeros=$P fixed shares"; HOST_CALLS_PRUNE=$P OPTY_HIP_SCATTER_SHARES=1 python tools/host_path_calls.py 40 2>&1 | tail -2 | head -1
done
