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

LLVM = '/opt/rocm/lib/llvm/bin'


def disassemble(path):
    if path.endswith('.s'):
        return open(path).read()
    with tempfile.TemporaryDirectory() as tmp:
        obj = os.path.join(tmp, 'd.o')
        subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'),
                        '--unbundle', '--type=o', '--input=' + path,
                        '--targets=hipv4-amdgcn-amd-amdhsa--gfx950',
                        '--output=' + obj], check=True, capture_output=True)
        return subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', obj],
                              check=True, capture_output=True,
                              text=True).stdout


def kernels(text):
    out, name = {}, None
    for ln in text.splitlines():
        m = re.match(r'^[0-9a-f]+ <(\w+)>:', ln)
        if m:
            name = m.group(1)
            out[name] = []
            continue
        m = re.match(r'\s*(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', ln)
        if m and name:
            out[name].append((int(m.group(3), 16), m.group(1), m.group(2)))
    return out


def _defs(op, args):
    """Vector registers an instruction writes (its first operand, for the
    instructions that have a vector destination)."""
    if not (op.startswith(('ds_read', 'buffer_load', 'global_load',
                           'scratch_load', 'flat_load')) or
            (op.startswith('v_') and not op.startswith(
                ('v_cmp', 'v_accvgpr_write', 'v_readlane',
                 'v_readfirstlane', 'v_nop')))):
        return []
    first = args.split(',')[0].strip()
    m = re.match(r'v\[(\d+):(\d+)\]$', first)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', first)
    return [int(m.group(1))] if m else []


def hazards(ins):
    """``[(agpr, vgpr, depth of its definition, copy address, copy depth,
    read address, read depth)]``: a vector register DEFINED under a wider
    EXEC than the one it is copied under, read back under a wider one."""
    depth = 0
    writes, reads, defined = {}, {}, {}
    for addr, op, args in ins:
        # if / else as hipcc lays it out: s_and_saveexec (then-arm: +1) ...
        # either  s_or_saveexec (EXEC = the whole region again: -1), s_xor
        # exec (else-arm: +1)  or  s_andn2_saveexec (then -> else, EXEC never
        # whole in between: 0) ... s_or_b64 exec (join: -1)
        if op.startswith('s_and_saveexec'):
            depth += 1
        elif op.startswith('s_or_saveexec'):
            depth = max(0, depth - 1)
        elif op == 's_xor_b64' and args.startswith('exec, exec'):
            depth += 1
        elif op == 's_or_b64' and args.startswith('exec, exec'):
            depth = max(0, depth - 1)
        m = re.match(r'a(\d+), v(\d+)', args) \
            if op == 'v_accvgpr_write_b32' else None
        if m:
            v = int(m.group(2))
            writes.setdefault(int(m.group(1)), []).append(
                (addr, depth, v, defined.get(v, 0)))
        m = re.match(r'v\d+, a(\d+)', args) if op == 'v_accvgpr_read_b32' \
            else None
        if m:
            reads.setdefault(int(m.group(1)), []).append((addr, depth))
        for v in _defs(op, args):
            defined[v] = depth
    out = []
    for a, ws in sorted(writes.items()):
        for waddr, wd, v, dd in ws:
            if dd >= wd:
                continue        # defined where it is copied: its own lanes
            # the reads this copy serves: up to the next copy into the register
            later = [x[0] for x in ws if x[0] > waddr]
            end = min(later) if later else float('inf')
            for raddr, rd in reads.get(a, []):
                if waddr < raddr < end and rd < wd:
                    out.append((a, v, dd, waddr, wd, raddr, rd))
                    break
    return out


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
