#!/usr/bin/env python
"""Developer tool (GPU box): does tools/o3_repro/reduced_biped_csr_persistent_
O2.hip.xz still show its fault?  Compiles it as recorded, runs the fused
kernel on the 8-node verification problem into a vector of NaNs and counts the
values of equation row 11 that were never stored."""
import json, lzma, os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
import opty_amd
from opty_amd import hip_backend as hb
from examples import problems
TAG = 'biped_csr_persistent_O2'
src = lzma.open(os.path.join(REPO, 'tools', 'o3_repro', 'reduced_%s.hip.xz' % TAG), 'rt').read()
info = json.load(open(os.path.join(REPO, 'tools', 'o3_repro', TAG + '.json')))
col = opty_amd.ConstraintCollocator(**info['collocator_kwargs'], **problems.build(info['problem']))
hsaco = hb.compile_module(src, col.tmp_dir, opt_level=info['opt_level'], extra_flags=tuple(info['extra_flags']))
res = hb.kernel_resources(hsaco)['opty_conjac']
N, free = col._verification_inputs()
ncn = N - 1
rs = list(col._build_program().row_start)
S, L = rs[11], rs[12] - rs[11]
desc = dict(col._descriptor(info['meta']), N=N, num_inst=0, nnz_inst=0, num_inst_atoms=0, inst_folded=0)
h = hb.HipProblem(desc, hsaco)
if not col._variable_duration:
    h.set_interval(col.node_time_interval)
if col.num_known_parameters:
    h.set_known_parameters(np.array([float(col.known_parameter_map[p]) for p in col.known_parameters]))
h.set_block_pattern(col._program.pattern)
for pattern in hb.POISONS:
    d = hb.DeviceVector(free); dj = hb.DeviceVector(np.full(h.nnz, np.nan)); dc = hb.DeviceVector(np.full(col.num_eom*ncn, np.nan))
    hb.poison_registers(pattern)
    h.eval_con_jac(d, dc, dj, hb.DEVICE); h.synchronize()
    jac = dj.numpy()
    print('%d lines; opty_conjac %d VGPRs, %d spilled, %d spilled SGPRs; registers %#x before the launch: %d of the %d values '
          'of row 11 (entries [%d, %d)) never stored; elsewhere %d'
          % (len(src.splitlines()), res['.vgpr_count'], res['.vgpr_spill_count'], res['.sgpr_spill_count'], pattern,
             int(np.isnan(jac[S*ncn:(S + L)*ncn]).sum()), L*ncn, S, S + L,
             int(np.isnan(jac).sum() - np.isnan(jac[S*ncn:(S + L)*ncn]).sum())))
