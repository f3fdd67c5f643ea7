cd $GRAFT_REPO_ROOT
cat /proc/loadavg
summ() { tail -1 /tmp/hp.out | python -c "
import sys, ast
d = ast.literal_eval(sys.stdin.read()); print('   jac %.2f (min %.2f)  pruned %.2f (min %.2f)  varying_first %.2f' % (d['jac'], d['jac_min'], d['jac_pruned'], d['jac_pruned_min'], d['jac_varying_first']))"; }
for i in 1 2 3; do for T in 16 8 4; do
  echo "threads $T"; OPTY_HIP_HOST_THREADS=$T python tools/bench_host_path_only.py > /tmp/hp.out 2>/dev/null; summ
done; done
cat /proc/loadavg
