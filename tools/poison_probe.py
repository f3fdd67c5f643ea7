#!/usr/bin/env python
"""Developer tool (GPU box): does a frozen miscompiling module read registers it
never wrote?  Before every evaluation of the module's Jacobian kernel a
poison kernel (tools/poison_kernel.hip: 4096 waves, every one fills all 256
VGPRs, 256 AGPRs and the free SGPRs with one 32-bit pattern) leaves a known
pattern in every SIMD's register file.  A correct kernel's output cannot
depend on it.

    hipcc --offload-arch=gfx950 -O1 --genco tools/poison_kernel.hip -o tools/poison_kernel.bin
    python tools/poison_probe.py <tag> [...]
"""
import ctypes
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np                                            # noqa: E402
import torch                                                  # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from examples import problems                                 # noqa: E402


def hip_runtime():
    hb.load_library()
    for line in open('/proc/self/maps'):
        if 'libamdhip64' in line:
            return ctypes.CDLL(line.split()[-1])
    raise RuntimeError('no HIP runtime mapped')


def main():
    import json
    import lzma
    torch.zeros(1).cuda()
    rt = hip_runtime()
    mod, fn = ctypes.c_void_p(), ctypes.c_void_p()
    path = os.path.join(REPO, 'tools', 'poison_kernel.bin').encode()
    assert rt.hipModuleLoad(ctypes.byref(mod), path) == 0
    assert rt.hipModuleGetFunction(ctypes.byref(fn), mod, b'opty_poison') == 0

    def poison(pattern):
        pat, sink = ctypes.c_uint(pattern), ctypes.c_void_p(0)
        args = (ctypes.c_void_p*2)(ctypes.addressof(pat),
                                   ctypes.addressof(sink))
        rc = rt.hipModuleLaunchKernel(fn, 4096, 1, 1, 64, 1, 1, 0, None, args,
                                      None)
        assert rc == 0, rc
        assert rt.hipDeviceSynchronize() == 0

    for tag in sys.argv[1:]:
        d = os.path.join(REPO, 'tools', 'o3_repro')
        with lzma.open(os.path.join(d, tag + '.hip.xz'), 'rt') as f:
            source = f.read()
        info = json.load(open(os.path.join(d, tag.replace('reduced_', '')
                                           + '.json')))
        kw = dict(info['collocator_kwargs'])
        if info.get('launch_nodes'):
            kw['launch_nodes'] = info['launch_nodes']
        col = opty_amd.ConstraintCollocator(
            **kw, **problems.build(info['problem']))
        meta = info['meta']
        hsaco = hb.compile_module(source, opt_level=info['opt_level'])
        rcon, rjac, con_row, jac_row = col._reference_values()
        N, free = col._verification_inputs()
        desc = dict(col._descriptor(meta), N=N, num_inst=0, nnz_inst=0,
                    num_inst_atoms=0, inst_folded=0)
        if col._jacobian_layout == 'varying_first':
            desc['layout'] = 0
        h = hb.HipProblem(desc, hsaco)
        if not col._variable_duration:
            h.set_interval(col.node_time_interval)
        if col.num_known_parameters:
            h.set_known_parameters(np.array(
                [float(col.known_parameter_map[p])
                 for p in col.known_parameters]))
        if col._program.pruned or col._jacobian_layout == 'csr':
            h.set_block_pattern(col._program.pattern)
        scale = np.zeros(int(jac_row.max()) + 1)
        np.maximum.at(scale, jac_row, np.abs(rjac))
        scale = np.maximum(scale, 1e-300)[jac_row]
        outs = []
        for pattern in (0x0, 0xffffffff, 0x7ff80000, 0x3ff00000, 0x0):
            poison(pattern)
            jac = np.full(h.nnz, np.nan)
            h.eval_jac(free, jac, hb.HOST)
            got = jac[:len(rjac)]
            with np.errstate(invalid='ignore'):
                bad = ~(np.abs(got - rjac)/scale <= 1e-9)
            outs.append(got.copy())
            print('%s  registers poisoned with 0x%08x: %d of %d Jacobian '
                  'values wrong (NaN %d), first wrong indices %s' % (
                      tag, pattern, int(bad.sum()), len(got),
                      int(np.isnan(got).sum()), np.where(bad)[0][:6]),
                  flush=True)
        same = all(np.array_equal(outs[0], o, equal_nan=True)
                   for o in outs[1:])
        print('%s: the kernel\'s output %s on what the registers held before '
              'it ran' % (tag, 'does NOT depend' if same else 'DEPENDS'),
              flush=True)
        h.close()


if __name__ == '__main__':
    main()
