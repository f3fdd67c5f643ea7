#!/usr/bin/env python
"""do the four frozen wrong builds (tools/o3_repro)
come out right when the register allocator is kept from parking vector
registers in accumulation registers -- and what does that cost?

Variants per module (same source, same -O level as recorded):
  recorded        as frozen (the wrong build)
  no_agpr_spill   -mllvm -amdgpu-spill-vgpr-to-agpr=0   (spills go to scratch)
  num_vgpr_256    __attribute__((amdgpu_num_vgpr(256))) on the Jacobian
                  kernels (the allocation cannot reach into the AGPR half)
  waves_per_eu_2  __attribute__((amdgpu_waves_per_eu(2, 2))): 256 unified
                  registers per wave, no accumulation half
  waves_2_no_agpr the last two together
Reported per variant and kernel: VGPR / AGPR counts, spilled VGPRs, scratch
bytes, copies the static ISA check finds (opty_amd.isa_check), and -- on a GPU
-- the referee's verdict (instruction tape, three register poisons).

    agpr_experiment.py --prebuild     (no GPU: compiles into the in-tree cache)
    agpr_experiment.py                (GPU box: verdicts, table on stdout)
"""
import os
import re
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

TAGS = ('one_legged_csr_O1', 'biped_20_strips_O2',
        'one_legged_park_spill_O2', 'biped_csr_persistent_O2')
NO_AGPR = ('-mllvm', '-amdgpu-spill-vgpr-to-agpr=0')


def with_num_vgpr(source, n=256):
    out, hits = re.subn(
        r'(__global__\s+void\s+)(__launch_bounds__\(\d+\)\s+)?'
        r'(opty_jac|opty_conjac)\b',
        lambda m: '%s%s__attribute__((amdgpu_num_vgpr(%d))) %s' % (
            m.group(1), m.group(2) or '', n, m.group(3)), source)
    assert hits >= 2, hits
    return out


def with_waves_per_eu(source, n=2):
    out, hits = re.subn(
        r'(__global__\s+void\s+)(__launch_bounds__\(\d+\)\s+)?'
        r'(opty_jac|opty_conjac)\b',
        lambda m: '%s%s__attribute__((amdgpu_waves_per_eu(%d, %d))) %s' % (
            m.group(1), m.group(2) or '', n, n, m.group(3)), source)
    assert hits >= 2, hits
    return out


def variants(source, info):
    base = tuple(info.get('extra_flags', ()))
    yield 'recorded', source, base
    yield 'no_agpr_spill', source, base + NO_AGPR
    yield 'num_vgpr_256', with_num_vgpr(source), base
    # two waves per SIMD: 256 unified registers per wave, all of them
    # addressable as v0-v255 -- the only switch found that takes the
    # accumulation registers away from the allocator
    yield 'waves_per_eu_2', with_waves_per_eu(source), base
    yield 'waves_2_no_agpr', with_waves_per_eu(source), base + NO_AGPR


def main():
    prebuild = '--prebuild' in sys.argv
    import test_hip_parity as thp
    from examples import problems
    import opty_amd
    from opty_amd import hip_backend as hb, isa_check
    print('# module / variant: kernel vgprs+agprs, spilled vgprs, scratch '
          'bytes, ISA copies%s' % ('' if prebuild else ' -> referee'))
    for tag in TAGS:
        source, info = thp.frozen_module(tag)
        kw = dict(info['collocator_kwargs'])
        if info.get('launch_nodes'):
            kw['launch_nodes'] = info['launch_nodes']
        col = None
        for label, src, flags in variants(source, info):
            try:
                hsaco = hb.compile_module(src, opt_level=info['opt_level'],
                                          extra_flags=flags)
            except ImportError as err:
                print('%-26s %-14s does not compile: %s' % (
                    tag, label, str(err).splitlines()[-1][:120]))
                continue
            res = hb.cached_kernel_resources(hsaco)
            copies = isa_check.exec_copies(hsaco)
            cells = []
            for k in ('opty_jac', 'opty_conjac'):
                r = res[k]
                cells.append('%s %d+%d v, %d spilled, %d B scratch, %d '
                             'copies' % (
                                 k[5:], r['.vgpr_count'],
                                 r.get('.agpr_count', 0),
                                 r['.vgpr_spill_count'],
                                 r['.private_segment_fixed_size'],
                                 copies.get(k, 0)))
            verdict = ''
            if not prebuild:
                if col is None:
                    col = opty_amd.ConstraintCollocator(
                        **kw, **problems.build(info['problem']))
                try:
                    v = col._verify_build(hsaco, info['meta'], force=True)
                    verdict = ' -> RIGHT (worst %.1e)' % v['worst']
                except hb.BuildRejected as err:
                    verdict = ' -> WRONG %s' % {
                        k: '%.2g' % e
                        for k, e in err.verdict['errors'].items()}
                # a verdict is cached next to the code object: remove it so
                # that another box judges afresh
                try:
                    os.remove(hsaco + '.crosscheck.json')
                except OSError:
                    pass
            print('%-26s %-14s %s%s' % (tag, label, '; '.join(cells),
                                        verdict), flush=True)


if __name__ == '__main__':
    main()
