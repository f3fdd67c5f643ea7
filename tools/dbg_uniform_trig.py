"""Developer tool (GPU box): the persistent row-sorted biped builds that hipcc
7.2 gets wrong (tools/o3_repro/biped_csr_persistent_O2), printed once more with
sincos behind a wave-uniform test (EmitOptions.fast_trig = 2): referee's
verdict and values against the default build."""
import sys, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, opty_amd
from opty_amd import hip_backend as hb
from examples import problems
for name in ('biped_mid_small', 'biped_small'):
    pkw = problems.build(name)
    col = opty_amd.ConstraintCollocator(jacobian_layout='csr', **pkw)
    free = problems.make_free(col.num_free, seed=11, variable_duration=col._variable_duration)
    j0 = np.array(col.generate_jacobian_function()(free)); c0 = col.generate_constraint_function()(free)
    for ft in (0, 2):
        opts = copy.copy(col._printer_options()); opts.order = opts.fused_order = 'list'; opts.fast_trig = ft
        sib = opty_amd.ConstraintCollocator(jacobian_layout='csr', emit_options=opts, **pkw)
        try:
            sib.hip
        except hb.BuildRejected as err:
            print('%s fast_trig=%d: REFUSED %s' % (name, ft, {k: '%.2g' % v for k, v in err.verdict['errors'].items()}), flush=True)
            continue
        j1 = np.array(sib.generate_jacobian_function()(free))
        c2, j2 = np.empty_like(c0), np.empty_like(j0)
        sib.hip.eval_con_jac(free, c2, j2, hb.HOST)
        s = np.abs(j0).max()
        print('%s fast_trig=%d: accepted (worst %.2g); against the default build: jac %.2e fused %.2e con %.2e of the scale'
              % (name, ft, sib._build_verdict['worst'], np.abs(j1 - j0).max()/s, np.abs(j2 - j0).max()/s,
                 np.abs(c2 - c0).max()/max(1.0, np.abs(c0).max())), flush=True)
