"""Measured launch plans (``opty_amd/launch_plan.py``): keys, lookup, the
candidate set around the printer's seeds, and that a recorded plan really
changes the module a collocator builds (CPU; the timing itself needs a GPU)."""
import json

import pytest

import opty_amd
from opty_amd import launch_plan as lp
from opty_amd.codegen.emit_hip import EmitOptions
from examples import problems


def test_buckets_and_keys():
    assert [lp.bucket(b) for b in (1, 98, 196, 391, 782, 1563, 6250,
                                   10**6)] == [0, 7, 8, 9, 10, 11, 12, 12]
    col = opty_amd.ConstraintCollocator(
        **problems.build('pend3_link_midpoint_small'))
    prog = col._build_program()
    key = lp.key_of(prog, 1563)
    sha, b, arch = key.split(':')
    assert (len(sha), b, arch) == (20, '11', 'gfx950')
    # the identity follows the equations, not the node count
    factory, fkw = problems.CONFIGS['pend3_link_midpoint_small']
    other = opty_amd.ConstraintCollocator(
        **factory(**dict(fkw, num_nodes=999)))
    assert lp.problem_sha(other._build_program()) == sha
    be = opty_amd.ConstraintCollocator(
        **factory(**dict(fkw, method='backward euler')))
    assert lp.problem_sha(be._build_program()) != sha


def test_candidates_surround_the_seed():
    col = opty_amd.ConstraintCollocator(
        **problems.build('config3_10link_small'))
    prog = col._build_program()
    cands, geo = lp.candidates(prog, 1563)
    assert cands[0][0] == 'seed' and geo['line_mode']
    fused = sorted(kw['fused_groups'] for _, kw in cands)
    jac = sorted(kw['groups'] for _, kw in cands)
    assert fused[0] < geo['fused'] < fused[-1]
    assert jac[0] < geo['jac'] < jac[-1]
    assert min(fused + jac) >= -(-3*geo['live']//5)
    # a small block has one choice: how its tile is flushed
    small = opty_amd.ConstraintCollocator(
        **problems.build('chaplygin_be_small'))
    cands, geo = lp.candidates(small._build_program(), 6250)
    assert [c[0] for c in cands] == ['seed', 'chunk'] and not geo['line_mode']


def test_a_recorded_plan_is_applied(tmp_path, monkeypatch):
    path = tmp_path/'plans.json'
    monkeypatch.setenv('OPTY_LAUNCH_PLANS', str(path))
    kw = problems.build('config3_10link_small')
    col = opty_amd.ConstraintCollocator(launch_nodes=99999, **kw)
    prog = col._build_program()
    _, meta = col.generate_source()
    seed = meta['geometry']
    assert lp.lookup(prog, 1563) is None                 # no file yet
    lp.record(lp.key_of(prog, 1563), dict(
        options=dict(groups=seed['jac'] + 2, fused_groups=seed['fused'] - 1),
        seed=dict(jac=seed['jac'], fused=seed['fused']), measured_ms={},
        nodes=99999, device='test'))
    assert json.loads(path.read_text())
    col2 = opty_amd.ConstraintCollocator(launch_nodes=99999, **kw)
    _, meta2 = col2.generate_source()
    assert meta2['kernels']['jac']['groups'] == seed['jac'] + 2
    assert meta2['kernels']['conjac']['groups'] == \
        seed['fused'] - 1 + seed['con_waves']
    # another launch size, explicit printer options and a disabled file are
    # not affected
    col3 = opty_amd.ConstraintCollocator(launch_nodes=12500, **kw)
    assert col3.generate_source()[1]['geometry']['occupancy'] == 2
    col4 = opty_amd.ConstraintCollocator(launch_nodes=99999,
                                         emit_options=EmitOptions(), **kw)
    assert col4.generate_source()[1]['sha'] == meta['sha']
    monkeypatch.setenv('OPTY_LAUNCH_PLANS', 'off')
    col5 = opty_amd.ConstraintCollocator(launch_nodes=99999, **kw)
    assert col5.generate_source()[1]['sha'] == meta['sha']
    # an entry written by another printer version is ignored
    monkeypatch.setenv('OPTY_LAUNCH_PLANS', str(path))
    lp.record(lp.key_of(prog, 1563), dict(options=dict(no_such_knob=1)))
    assert lp.lookup(prog, 1563) is None


def test_the_tracked_plan_file_is_well_formed():
    import os
    if not os.path.exists(lp.DEFAULT_FILE):
        pytest.skip('no plans recorded yet')
    with open(lp.DEFAULT_FILE) as f:
        plans = json.load(f)
    for key, entry in plans.items():
        sha, b, arch = key.split(':')
        assert arch == 'gfx950' and 0 <= int(b) <= 12 and len(sha) == 20
        EmitOptions(**entry['options'])
        if entry.get('pinned') is not None:
            # a build that replaced one the verification refused
            assert set(entry) >= {'options', 'pinned', 'refused', 'nodes'}
            assert set(entry['pinned']) >= {'label', 'replaces'}
            continue
        assert set(entry) >= {'options', 'seed', 'measured_ms', 'nodes',
                              'device'}
