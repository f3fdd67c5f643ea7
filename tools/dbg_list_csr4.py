"""Developer tool (GPU box): are the wrong values of tools/o3_repro/
biped_csr_persistent_O2 stored wrong or never stored?  The fused kernel into
device vectors that hold a sentinel, register files poisoned with zeros."""
import sys, json, lzma
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, opty_amd
from opty_amd import hip_backend as hb
from examples import problems
tag = 'biped_csr_persistent_O2'
src = lzma.open('/root/repo/tools/o3_repro/%s.hip.xz' % tag, 'rt').read()
info = json.load(open('/root/repo/tools/o3_repro/%s.json' % tag))
col = opty_amd.ConstraintCollocator(**info['collocator_kwargs'], **problems.build(info['problem']))
hsaco = hb.compile_module(src, col.tmp_dir, opt_level=info['opt_level'], extra_flags=tuple(info['extra_flags']))
meta = info['meta']
rcon, rjac, con_row, jac_row = col._reference_values()
N, free = col._verification_inputs()
desc = dict(col._descriptor(meta), N=N, num_inst=0, nnz_inst=0, num_inst_atoms=0, inst_folded=0)
h = hb.HipProblem(desc, hsaco)
if not col._variable_duration:
    h.set_interval(col.node_time_interval)
if col.num_known_parameters:
    h.set_known_parameters(np.array([float(col.known_parameter_map[p]) for p in col.known_parameters]))
h.set_block_pattern(col._program.pattern)
for pattern in (0x0, 0x7ff80000):
    for sentinel in (12345.0, -777.0):
        d = hb.DeviceVector(free); dc = hb.DeviceVector(np.full(len(rcon), sentinel)); dj = hb.DeviceVector(np.full(h.nnz, sentinel))
        hb.poison_registers(pattern)
        h.eval_con_jac(d, dc, dj, hb.DEVICE); h.synchronize()
        j = dj.numpy()
        bad = np.flatnonzero(~(np.abs(j - rjac) <= 1e-9*np.abs(rjac).max()))
        vals, counts = np.unique(j[bad], return_counts=True)
        print('poison %#x sentinel %g: %d wrong; their values: %s' % (pattern, sentinel, len(bad), dict(zip(vals[:6].tolist(), counts[:6].tolist()))))
