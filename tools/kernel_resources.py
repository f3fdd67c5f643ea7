#!/usr/bin/env python
"""Developer tool: VGPR / spill / LDS figures of every kernel in a gfx950 code
object (from the AMDGPU metadata note).  Usage: kernel_resources.py x.hsaco ..."""
import re
import subprocess
import sys
import tempfile

READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
KEYS = ('.vgpr_count', '.agpr_count', '.sgpr_count', '.vgpr_spill_count',
        '.sgpr_spill_count', '.private_segment_fixed_size',
        '.group_segment_fixed_size')


BUNDLER = '/opt/rocm/lib/llvm/bin/clang-offload-bundler'


def resources(path):
    # `hipcc --genco` writes an offload bundle; take the gfx950 ELF out of it
    with tempfile.NamedTemporaryFile(suffix='.elf') as elf:
        subprocess.run([BUNDLER, '--unbundle', '--type=o', '--input=' + path,
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        '--output=' + elf.name], check=True)
        txt = subprocess.run([READELF, '--notes', elf.name],
                             capture_output=True, text=True).stdout
    out = {}
    for blk in txt.split('- .agpr_count')[1:]:
        blk = '.agpr_count' + blk
        name = re.search(r'\.name:\s+(\S+)', blk).group(1)
        out[name] = {k: int(re.search(re.escape(k) + r':\s+(\d+)', blk)
                            .group(1)) for k in KEYS}
    return out


if __name__ == '__main__':
    for path in sys.argv[1:]:
        print(path)
        for name, r in sorted(resources(path).items()):
            print('  %-12s vgpr %3d agpr %3d sgpr %3d  spills v %4d s %5d  '
                  'scratch %4d B  lds %6d B' % (
                      name, r['.vgpr_count'], r['.agpr_count'],
                      r['.sgpr_count'], r['.vgpr_spill_count'],
                      r['.sgpr_spill_count'],
                      r['.private_segment_fixed_size'],
                      r['.group_segment_fixed_size']))
