"""Which kernels serve the entry points (``opty_hip_desc.routing``): measured
by the handle on its own device, never a kernel the build banned."""
import numpy as np
import pytest

from examples import problems

pytestmark = pytest.mark.gpu


def _reference(kw):
    import opty_amd
    ref = opty_amd.ConstraintCollocator(**kw)
    free = problems.make_free(ref.num_free, seed=4)
    return free, ref.generate_constraint_function()(free), \
        np.array(ref.generate_jacobian_function()(free))


def test_handle_calibrates_its_routing_on_first_use(monkeypatch):
    """``OPTY_HIP_ROUTE_CALIBRATE``: before the first Jacobian launch of a
    size the plan's flags are in force; the first ``EVAL_FUSED`` /
    ``EVAL_JAC`` launch measures opty_conjac, opty_con and opty_jac on the
    device and the entry points launch the faster candidates from then on
    (``opty_hip_routing``); values are those of the same expressions
    whichever kernel serves."""
    import opty_amd
    from opty_amd import hip_backend as hb
    kw = problems.build('config3_10link_small')
    free, con0, jac0 = _reference(kw)
    col = opty_amd.ConstraintCollocator(**kw)
    hip = col.hip
    assert hip.desc['routing'] & hb.ROUTE_CALIBRATE
    before = hip.routing()
    assert before['routing'] == 'plan' and 'ms' not in before
    con, jac = np.empty_like(con0), np.empty_like(jac0)
    hip.eval_con_jac(free, con, jac, hb.HOST)
    after = hip.routing()
    assert after['routing'] == 'calibrated'
    ms = after['ms']
    assert all(0.0 < ms[k] < 50.0 for k in ('opty_conjac', 'opty_con',
                                            'opty_jac')), ms
    # the decision is the measured one (1 % + 0.3 us in favour of the plan)
    pair = ms['opty_con'] + ms['opty_jac']
    if after['fused_loses']:
        assert pair <= ms['opty_conjac']*1.0001
    else:
        assert ms['opty_conjac'] <= pair*1.0101 + 3.1e-4
    if after['jac_via_fused']:
        assert ms['opty_conjac'] <= ms['opty_jac']*1.0001
    # other launch sizes are measured on their own
    assert hip.routing(nodes=100000)['routing'] == 'plan'
    jac2 = np.empty_like(jac0)
    hip.eval_jac(free, jac2, hb.HOST)
    for got, want in ((con, con0), (jac, jac0), (jac2, jac0)):
        np.testing.assert_allclose(got, want, rtol=1e-12,
                                   atol=1e-12*np.abs(want).max())
    hip.close()
    # OPTY_HIP_ROUTING=plan: the plan's flags as they are
    monkeypatch.setenv('OPTY_HIP_ROUTING', 'plan')
    col2 = opty_amd.ConstraintCollocator(**kw)
    col2.hip.eval_con_jac(free, con, jac, hb.HOST)
    assert col2.hip.routing()['routing'] == 'plan'
    col2.hip.close()


@pytest.mark.parametrize('banned', ['opty_jac', 'opty_conjac'])
def test_banned_kernel_is_never_launched(banned, monkeypatch, tmp_path):
    """A kernel the build marked unusable (it spills vector registers
    whatever the cut) is replaced by an EMPTY kernel here: every entry point
    still returns the right values, because none launches it."""
    import re
    import opty_amd
    from opty_amd import hip_backend as hb
    kw = problems.build('pend3_link_midpoint_small')
    free, con0, jac0 = _reference(kw)
    col = opty_amd.ConstraintCollocator(tmp_dir=str(tmp_path),
                                        verify_builds='off', **kw)
    source, meta = col.generate_source()
    # body of the banned kernel -> nothing
    m = re.search(r'^%s\s*\([^)]*\)\s*\{' % banned, source, re.M)
    assert m, 'kernel not found in the printed module'
    depth, k = 1, m.end()
    while depth:
        depth += {'{': 1, '}': -1}.get(source[k], 0)
        k += 1
    gutted = source[:m.end()] + '}' + source[k:]
    hsaco = hb.compile_module(gutted, str(tmp_path))
    meta = dict(meta, banned_kernels=[banned])
    monkeypatch.setattr(col, '_build_code_object',
                        lambda opt_level=None: (hsaco, meta))
    hip = col.hip
    bit = hb.ROUTE_NO_JAC_KERNEL if banned == 'opty_jac' \
        else hb.ROUTE_NO_FUSED_KERNEL
    assert hip.desc['routing'] & bit
    r = hip.routing()
    assert r['fused_loses'] == (banned == 'opty_conjac')
    assert r['jac_via_fused'] == (banned == 'opty_jac')
    con, jac, jac2 = np.full_like(con0, np.nan), \
        np.full_like(jac0, np.nan), np.full_like(jac0, np.nan)
    hip.eval_con_jac(free, con, jac, hb.HOST)
    hip.eval_jac(free, jac2, hb.HOST)
    con2 = np.full_like(con0, np.nan)
    hip.eval_con(free, con2, hb.HOST)
    for got, want in ((con, con0), (con2, con0), (jac, jac0), (jac2, jac0)):
        np.testing.assert_allclose(got, want, rtol=1e-12,
                                   atol=1e-12*np.abs(want).max())
    if banned == 'opty_conjac':
        dfree = hb.DeviceVector(free)
        dcon, djac = hb.DeviceVector(con0), hb.DeviceVector(jac0)
        with pytest.raises(hb.HipBackendError, match='unusable'):
            hip.time_eval(hb.EVAL_FUSED_KERNEL, dfree, dcon, djac, 1)
        for v in (dfree, dcon, djac):
            v.close()
    hip.close()
