"""GPU tool: a few jacobian(free) calls of config 3 through the host path with
OPTY_HIP_TRACE=1 (where the time of a call goes: stderr)."""
import os, sys, time
os.environ['OPTY_HIP_TRACE']='1'
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), '..')))
import numpy as np
import opty_amd
from examples import problems
kw=problems.build('config3_10link')
col=opty_amd.ConstraintCollocator(**kw)
jf=col.generate_jacobian_function()
frees=[problems.make_free(col.num_free, seed=s) for s in range(3)]
for k in range(20):
    t0=time.perf_counter(); jf(frees[k%3]); print('call %d %.3f ms'%(k,1e3*(time.perf_counter()-t0)), file=sys.stderr)
# the pipeline's vector against the whole-vector copy of the same evaluation
from opty_amd import hip_backend as hb
ref = hb.pinned_empty(col.hip.nnz)
col.hip.eval_jac(frees[19 % 3], ref, hb.HOST)
got = np.array(jf(frees[19 % 3]))
worst = float(np.max(np.abs(got - ref)/np.maximum(1e-300, np.maximum(np.abs(ref), 1.0))))
print('pipeline vs dense copy: worst difference %.3g (%d values)' % (worst, got.size), file=sys.stderr)
assert worst < 1e-12, worst
