cd $GRAFT_REPO_ROOT
cat /proc/loadavg
for i in 1 2 3; do
  python tools/bench_host_path_only.py 2>/dev/null | tail -1 | cut -c1-330
  OPTY_HIP_SCATTER_SLICES=1 python tools/bench_host_path_only.py 2>/dev/null | tail -1 | cut -c1-330
done
