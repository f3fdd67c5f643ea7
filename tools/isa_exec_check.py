#!/usr/bin/env python
"""Developer tool (CPU): lane-level check of a generated kernel's ISA -- is a
vector register copied into an accumulation register (the register
allocator's cheap spill: ``v_accvgpr_write_b32 aN, vM``) while EXEC is
narrowed, and read back (``v_accvgpr_read_b32 vK, aN``) under a wider EXEC?
Then the lanes that were switched off at the copy read whatever the register
file held: the hipcc 7.2 fault of DESIGN.md 4.1 (found in
tools/o3_repro/reduced_biped_csr_persistent_O2: the copy of ``lane`` sits in
the flow block between the two arms of an inlined sincos's argument-range
branch, executed with the mask of the first arm -- or with an empty one).

    isa_exec_check.py <code object | disassembly .s> [kernel ...]

The depth of EXEC narrowing is followed along the layout order of the
structured control flow hipcc emits (see ``hazards``).  Reported: every
copy of a vector register that was DEFINED under a wider EXEC than the copy
runs under and is read back under a wider one too (a value defined inside
the region, copied there and merged later is what a branch normally does)."""
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                '..'))
from opty_amd.isa_check import disassemble, kernels, hazards  # noqa: E402


def main():
    text = disassemble(sys.argv[1])
    ks = kernels(text)
    names = sys.argv[2:] or [k for k in ks if k.startswith('opty_')]
    total = 0
    for name in names:
        hz = hazards(ks[name])
        total += len(hz)
        for a, v, dd, waddr, wd, raddr, rd in hz[:8]:
            print('%s: v%d (defined with EXEC narrowed %d deep) copied to '
                  'a%d at %#x with EXEC narrowed %d deep, read back at %#x '
                  'with EXEC narrowed %d deep' % (name, v, dd, a, waddr, wd,
                                                  raddr, rd))
        if len(hz) > 8:
            print('%s: ... %d in all' % (name, len(hz)))
    print('%s: %d accumulation registers copied under a narrower EXEC than '
          'they are read under' % (os.path.basename(sys.argv[1]), total))
    return total


if __name__ == '__main__':
    sys.exit(1 if main() else 0)
