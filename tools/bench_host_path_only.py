"""Developer tool (GPU box): bench.py's host_path_ms block alone."""
import sys, os
sys.path.insert(0, '/root/repo')
import bench
from examples import problems
out = bench.host_path(problems.build('config3_10link'), 0)
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items() if k.startswith(('jac', 'pair'))})
