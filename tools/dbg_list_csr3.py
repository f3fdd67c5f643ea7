import sys, copy
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, opty_amd
from opty_amd import hip_backend as hb
from examples import problems
name = sys.argv[1]
pkw = problems.build(name)
col = opty_amd.ConstraintCollocator(jacobian_layout='csr', **pkw)
opts = copy.copy(col._printer_options())
opts.order = opts.fused_order = 'list'
sib = opty_amd.ConstraintCollocator(jacobian_layout='csr', emit_options=opts, verify_builds='off', **pkw)
hsaco, meta = sib._build_code_object()
ref = opty_amd.ConstraintCollocator(jacobian_layout='csr', emit_options=opts, **pkw)
try:
    print('referee:', ref._verify_build(hsaco, meta, force=True)['errors'])
except hb.BuildRejected as e:
    print('referee refuses:', e.verdict['errors'])
def wrong(free, tag):
    j0 = np.array(col.generate_jacobian_function()(free)); c0 = col.generate_constraint_function()(free)
    c2, j2 = np.empty_like(c0), np.empty_like(j0)
    sib.hip.eval_con_jac(free, c2, j2, hb.HOST)
    j1 = np.array(sib.generate_jacobian_function()(free))
    s = np.abs(j0).max()
    print(tag, 'fused wrong', int((np.abs(j2-j0) > 1e-9*s).sum()), 'jac wrong', int((np.abs(j1-j0) > 1e-9*s).sum()), 'nan in ref', int(np.isnan(j0).sum()))
wrong(problems.make_free(col.num_free, seed=11, variable_duration=col._variable_duration), 'make_free 11')
N, f = sib._verification_inputs(7, (-1.0, 1.0))
print('N', N, len(f), col.num_free)
wrong(f, 'referee inputs (-1,1)')
wrong(sib._verification_inputs(7, (0.1, 0.9))[1], 'referee inputs (0.1,0.9)')
# shard launches of the production handle: no instance tails
import torch
dev = torch.device('cuda:0'); hip = sib.hip; hip.use_torch_stream()
free = problems.make_free(col.num_free, seed=11)
j0 = np.array(col.generate_jacobian_function()(free))
ncn, P, M = N - 1, hip.desc['P'], hip.desc['M']
free = problems.make_free(col.num_free, seed=11, variable_duration=col._variable_duration)
j0 = np.array(col.generate_jacobian_function()(free)); c0 = col.generate_constraint_function()(free)
s = np.abs(j0).max()
for tag, over in (('as built', {}), ('no instance tails', dict(num_inst=0, nnz_inst=0, num_inst_atoms=0, inst_folded=0)),
                  ('tails by opty_inst', dict(inst_folded=0))):
    desc = dict(sib._descriptor(meta), **over)
    h = hb.HipProblem(desc, hsaco)
    sib2 = sib
    if not sib._variable_duration: h.set_interval(sib.node_time_interval)
    if sib.num_known_parameters:
        h.set_known_parameters(np.array([float(sib.known_parameter_map[p]) for p in sib.known_parameters]))
    h.set_block_pattern(sib._program.pattern)
    if desc['num_inst']:
        idx = sib.instance_constraints_free_index_map
        h.set_instance_indices([idx[f] for f in sib._inst_atoms], sib._inst_rows, sib._inst_cols)
    c2 = np.empty(h.num_constraints if hasattr(h, 'num_constraints') else len(c0) if desc['num_inst'] else M*ncn)
    j2 = np.empty(h.nnz)
    h.eval_con_jac(free, c2, j2, hb.HOST)
    n = min(len(j2), P*ncn)
    print(tag, 'fused wrong', int((np.abs(j2[:n] - j0[:n]) > 1e-9*s).sum()))
    h.close()
