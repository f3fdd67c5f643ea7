import sys, os
sys.path.insert(0, '.')
import numpy as np
import opty_amd
from opty_amd import hip_backend as hb
from examples import problems
name = sys.argv[1]; layout = sys.argv[2]
kw = problems.build(name)
os.environ['OPTY_CROSS_CHECK'] = 'off'
col = opty_amd.ConstraintCollocator(jacobian_layout=layout, **kw)
hsaco, meta = col._build_code_object()
src = col._built_source
twin = hb.compile_module(src, col.tmp_dir, opt_level='-O1')
print('hot', hb.high_pressure_kernels(hsaco), 'spills', hb.vgpr_spills(hsaco))
N = min(col.num_collocation_nodes, 131)
n, q = col.num_states, col.num_unknown_input_trajectories
rng = np.random.default_rng(7)
free = rng.uniform(-1, 1, (n+q)*N + col.num_unknown_parameters + int(col._variable_duration))
if col._variable_duration: free[-1] = 0.01
desc = dict(col._descriptor(meta), N=N, num_inst=0, nnz_inst=0, num_inst_atoms=0, inst_folded=0)
outs = {}
for tag, h_ in (('O2', hsaco), ('O1', twin)):
    for rep in range(2):
        h = hb.HipProblem(desc, h_)
        if not col._variable_duration: h.set_interval(col.node_time_interval)
        if col.num_known_parameters:
            h.set_known_parameters(np.array([float(col.known_parameter_map[p]) for p in col.known_parameters]))
        if col.num_known_input_trajectories:
            h.set_known_trajectories(np.ascontiguousarray(col._known_trajectory_array(np.ones(col.num_free))[:, :N]))
        if col._program.pruned or layout == 'csr':
            h.set_block_pattern(col._program.pattern)
        con = np.full(col.num_eom*(N-1), np.nan); jac = np.full(h.nnz, np.nan)
        h.eval_con(free, con, hb.HOST); h.eval_jac(free, jac, hb.HOST)
        con2 = np.full_like(con, np.nan); jac2 = np.full_like(jac, np.nan)
        h.eval_con_jac(free, con2, jac2, hb.HOST)
        outs[tag, rep] = (con, jac, con2, jac2)
        h.close()
names = ('con', 'jac', 'fused con', 'fused jac')
def diff(a, b):
    return [float(np.nanmax(np.abs(x-y))/max(np.nanmax(np.abs(x)),1e-300)) if np.isfinite(x).all() and np.isfinite(y).all() else 'nan:%d/%d'%(np.isnan(x).sum(), np.isnan(y).sum()) for x, y in zip(a, b)]
print('O2 rep0 vs rep1', diff(outs['O2',0], outs['O2',1]))
print('O1 rep0 vs rep1', diff(outs['O1',0], outs['O1',1]))
print('O2 vs O1       ', diff(outs['O2',0], outs['O1',0]))
print('O2 sep vs fused', diff(outs['O2',0][:2], outs['O2',0][2:]))
print('O1 sep vs fused', diff(outs['O1',0][:2], outs['O1',0][2:]))
# coo reference of the same small problem
col0 = opty_amd.ConstraintCollocator(**kw)
h0, m0 = col0._build_code_object()
d0 = dict(col0._descriptor(m0), N=N, num_inst=0, nnz_inst=0, num_inst_atoms=0, inst_folded=0)
h = hb.HipProblem(d0, h0)
if not col._variable_duration: h.set_interval(col.node_time_interval)
if col.num_known_parameters:
    h.set_known_parameters(np.array([float(col.known_parameter_map[p]) for p in col.known_parameters]))
if col.num_known_input_trajectories:
    h.set_known_trajectories(np.ascontiguousarray(col._known_trajectory_array(np.ones(col.num_free))[:, :N]))
con = np.empty(col.num_eom*(N-1)); jac = np.empty(h.nnz)
h.eval_con(free, con, hb.HOST); h.eval_jac(free, jac, hb.HOST)
print('coo con vs O2 con', diff([con],[outs['O2',0][0]]), 'vs O1 con', diff([con],[outs['O1',0][0]]))
print('sorted |jac| coo vs O2', diff([np.sort(np.abs(jac))],[np.sort(np.abs(outs['O2',0][1]))]), 'vs O1', diff([np.sort(np.abs(jac))],[np.sort(np.abs(outs['O1',0][1]))]))
