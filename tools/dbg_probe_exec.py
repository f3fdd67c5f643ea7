import json, lzma, os, sys
REPO='/root/repo'
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, 'tests'))
import numpy as np
import opty_amd
from opty_amd import hip_backend as hb
from examples import problems
TAG='biped_csr_persistent_O2'
src=open(os.path.join(REPO,'tools','o3_repro','probe_exec.hip')).read()
info=json.load(open(os.path.join(REPO,'tools','o3_repro',TAG+'.json')))
col=opty_amd.ConstraintCollocator(**info['collocator_kwargs'], **problems.build(info['problem']))
hsaco=hb.compile_module(src, col.tmp_dir, opt_level=info['opt_level'], extra_flags=tuple(info['extra_flags']))
N, free = col._verification_inputs(); ncn=N-1
rs=list(col._build_program().row_start); S,L=rs[11],rs[12]-rs[11]
desc=dict(col._descriptor(info['meta']), N=N, num_inst=0, nnz_inst=0, num_inst_atoms=0, inst_folded=0)
h=hb.HipProblem(desc, hsaco)
h.set_known_parameters(np.array([float(col.known_parameter_map[p]) for p in col.known_parameters]))
h.set_block_pattern(col._program.pattern)
for pattern in hb.POISONS:
    d=hb.DeviceVector(free); dj=hb.DeviceVector(np.full(h.nnz,np.nan)); dc=hb.DeviceVector(np.full(col.num_eom*ncn,np.nan))
    hb.poison_registers(pattern); h.eval_con_jac(d,dc,dj,hb.DEVICE); h.synchronize()
    jac=dj.numpy(); c=dc.numpy()
    print('poison %#x: row 11 never stored: %d of %d; at the flush: active lanes %s, exec lo %s hi %s, nvalid %s ncn %s nloc %s'
          % (pattern, int(np.isnan(jac[S*ncn:(S+L)*ncn]).sum()), L*ncn, c[0], hex(int(c[1])) if c[1]==c[1] else c[1], hex(int(c[2])) if c[2]==c[2] else c[2], c[3], c[4], c[5]))
