"""Static check of a generated kernel's ISA for the one hipcc 7.2 fault that is
understood at the instruction (profiles/r05_exec_fault.txt):
a vector register copied into an accumulation register (the register
allocator's cheap spill, ``v_accvgpr_write_b32 aN, vM``) while EXEC is
narrowed, and read back (``v_accvgpr_read_b32 vK, aN``) under a wider EXEC --
the lanes that were switched off at the copy read whatever the register file
held.

The depth of EXEC narrowing is followed along the layout order of the
structured control flow hipcc emits (``hazards``).  Reported: every copy of a
vector register that was DEFINED under a wider EXEC than the copy runs under
and is read back under a wider one too (a value defined inside the region,
copied there and merged later is what a branch normally does).

This is a BUILD GATE since r06 (``ConstraintCollocator._build_code_object``):
a code object with such a copy is replaced, before any GPU time is spent on
it, by the same geometry printed with ``fast_trig=2`` (sincos behind a
wave-uniform test: no EXEC-narrowing if / else left on the hot path) when
that sibling is clean; the count is part of the build's verdict either way.
It is a filter, not a proof -- the harmless instances of the pattern (lanes
beyond the last node, inside ``if (valid)``) look the same --; the referee
(the instruction tape) stays the judge of every build.

``tools/isa_exec_check.py`` is the command-line front end.
"""
import json
import os
import re
import subprocess
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


def exec_copies(hsaco_path, names=('opty_con', 'opty_jac', 'opty_conjac')):
    """``{kernel: count}`` of the copies :func:`hazards` reports in the
    kernels ``names`` of a code object (only kernels with at least one);
    cached next to the code object (``<hsaco>.isa.json``)."""
    side = hsaco_path + '.isa.json'
    try:
        with open(side) as f:
            got = json.load(f)
        if got.get('version') == 1:
            return {k: v for k, v in got['copies'].items() if k in names}
    except (OSError, ValueError, KeyError):
        pass
    ks = kernels(disassemble(hsaco_path))
    copies = {}
    for name, ins in ks.items():
        if name.startswith('opty_'):
            n = len(hazards(ins))
            if n:
                copies[name] = n
    try:
        tmp = side + '.%d.tmp' % os.getpid()
        with open(tmp, 'w') as f:
            json.dump(dict(version=1, copies=copies), f)
        os.replace(tmp, side)
    except OSError:
        pass
    return {k: v for k, v in copies.items() if k in names}


def noted(hsaco_path, key):
    """A remark stored next to a code object's verdict (``note``), or None."""
    try:
        with open(hsaco_path + '.isa.json') as f:
            return json.load(f).get('notes', {}).get(key)
    except (OSError, ValueError):
        return None


def note(hsaco_path, key, value):
    """Stores a remark next to a code object's verdict -- e.g. that its
    uniform-sincos sibling was built and found no better, so that the next
    build does not spend minutes of hipcc on finding out again."""
    side = hsaco_path + '.isa.json'
    try:
        with open(side) as f:
            got = json.load(f)
        got.setdefault('notes', {})[key] = value
        tmp = side + '.%d.tmp' % os.getpid()
        with open(tmp, 'w') as f:
            json.dump(got, f)
        os.replace(tmp, side)
    except (OSError, ValueError):
        pass

