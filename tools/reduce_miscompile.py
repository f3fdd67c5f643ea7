#!/usr/bin/env python
"""Developer tool (GPU box): shrinks a frozen miscompiling module
(``tools/o3_repro/<tag>.hip.xz``) with the instruction tape as the
interestingness test -- the compiler fault shows only in values computed on a
GPU, so the loop (edit, hipcc, run, compare) runs where the GPU is.

    python tools/reduce_miscompile.py <tag> [kernel]      # default: opty_jac

Stages (each candidate is compiled with the recorded switches, its Jacobian
kernel run on the seeded verification inputs, the entries of the kept strip
compared with the expression DAG evaluated by ``opty_hip_tape_run``):

 A. one kernel, one wave: the other kernels become empty shells (the runtime
    looks their names up), of the faulty kernel's ``switch`` only the strip
    whose entries are wrong stays;
 B. entries: ring writes of entries that come out RIGHT store 0.0 instead,
    what only they needed is deleted (dead-code elimination on the printed
    straight-line code); then halves of the wrong entries, as long as one
    wrong entry is left;
 C. chunks: flushes of chunks without a wrong entry are kept (they are the
    store path) but their values are constants.

Writes ``gpurun_out/reduced_<tag>.hip`` (self-contained: device header
inlined) and ``gpurun_out/reduced_<tag>.txt``: lines, registers, expected /
actual of the entries still wrong at the first nodes."""
import json
import lzma
import os
import re
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))

import numpy as np                                            # noqa: E402
import opty_amd                                               # noqa: E402
from opty_amd import hip_backend as hb                        # noqa: E402
from examples import problems                                 # noqa: E402

DEF = re.compile(r'^\s*const double (\w+) = (.*);\s*$')
PAIR = re.compile(r'^\s*double (\w+), (\w+); \w*sincos\((.*), &\w+, &\w+\);\s*$')
RING = re.compile(r'^(\s*ring\[(\d+) \+ lane\] = )(.*);\s*$')
WORD = re.compile(r'[A-Za-z_]\w*')


def kernels(lines):
    """``{name: (first line, end line)}`` of the module's kernels."""
    starts = [i for i, ln in enumerate(lines)
              if ln.startswith('extern "C" __global__')]
    out = {}
    for k, i in enumerate(starts):
        name = lines[i + 1].split('(')[0]
        # the kernel ends at the closing brace in column 0
        j = i + 2
        while lines[j] != '}':
            j += 1
        out[name] = (i, j + 1)
    return out


def shell(lines, i, j):
    """The kernel ``lines[i:j]`` as an empty body."""
    return lines[i:i + 2] + ['{', '}']


def dce(body):
    """Removes definitions nobody reads (iterated)."""
    while True:
        used = {}
        for ln in body:
            m = DEF.match(ln) or PAIR.match(ln)
            text = (m.group(2) if DEF.match(ln) else m.group(3)) if m else ln
            for w in WORD.findall(text):
                used[w] = used.get(w, 0) + 1
        keep, changed = [], False
        for ln in body:
            m = DEF.match(ln)
            if m and not used.get(m.group(1)):
                changed = True
                continue
            m = PAIR.match(ln)
            if m and not used.get(m.group(1)) and not used.get(m.group(2)):
                changed = True
                continue
            keep.append(ln)
        body = keep
        if not changed:
            return body


def main():
    tag = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else 'opty_jac'
    with lzma.open(os.path.join(REPO, 'tools', 'o3_repro', tag + '.hip.xz'),
                   'rt') as f:
        source = f.read()
    with open(os.path.join(REPO, 'tools', 'o3_repro', tag + '.json')) as f:
        info = json.load(f)
    kw = dict(info['collocator_kwargs'])
    if info.get('launch_nodes'):
        kw['launch_nodes'] = info['launch_nodes']
    col = opty_amd.ConstraintCollocator(**kw,
                                        **problems.build(info['problem']))
    meta = info['meta']
    csr = col._jacobian_layout == 'csr'
    flags = tuple(info.get('extra_flags', ()))
    P = meta['P']
    rcon, rjac, con_row, jac_row = col._reference_values()
    N, free = col._verification_inputs()
    ncn = N - 1
    # to_block[nd, e] = index of entry e of node nd in the kernels' vector
    if csr:
        # jac[S_j*ncn + i*L_j + pos] (row-sorted)
        rs = list(col._build_program().row_start)
        to_block = np.empty((ncn, P), dtype=np.int64)
        for j in range(len(rs) - 1):
            S, L = rs[j], rs[j + 1] - rs[j]
            to_block[:, S:S + L] = S*ncn + np.arange(ncn)[:, None]*L + \
                np.arange(L)[None, :]
    else:
        to_block = np.arange(ncn*P).reshape(ncn, P)
    want = rjac[to_block]
    scale = np.zeros(int(jac_row.max()) + 1)
    np.maximum.at(scale, jac_row, np.abs(rjac))
    ent_scale = np.maximum(scale[jac_row[to_block][0]], 1e-300)
    tmp = os.path.join(REPO, 'gpurun_out', 'reduce_cache')
    os.makedirs(tmp, exist_ok=True)
    tried = [0]

    def run(text):
        """Per-entry error (relative to the entry's row) of the module's
        Jacobian kernel on the verification problem, ``(P,)``."""
        tried[0] += 1
        hsaco = hb.compile_module(text, tmp, opt_level=info['opt_level'],
                                  extra_flags=flags)
        desc = dict(col._descriptor(meta), N=N, num_inst=0, nnz_inst=0,
                    num_inst_atoms=0, inst_folded=0)
        h = hb.HipProblem(desc, hsaco)
        try:
            if not col._variable_duration:
                h.set_interval(col.node_time_interval)
            if col.num_known_parameters:
                h.set_known_parameters(np.array(
                    [float(col.known_parameter_map[p])
                     for p in col.known_parameters]))
            if col.num_known_input_trajectories:
                h.set_known_trajectories(np.ascontiguousarray(
                    col._known_trajectory_array(
                        np.ones(col.num_free))[:, :N]))
            if col._program.pruned or csr:
                h.set_block_pattern(col._program.pattern)
            # device vectors that start as NaNs: a store that never happens
            # must not find an earlier kernel's value in a staging buffer
            dfree = hb.DeviceVector(free)
            djac = hb.DeviceVector(np.full(h.nnz, np.nan))
            dcon = hb.DeviceVector(np.full(col.num_eom*ncn, np.nan))
            hb.poison_registers(0)
            if kernel == 'opty_conjac':
                h.eval_con_jac(dfree, dcon, djac, hb.DEVICE)
            else:
                h.eval_jac(dfree, djac, hb.DEVICE)
            h.synchronize()
            jac = djac.numpy()
            for d in (dfree, djac, dcon):
                d.close()
            got = jac[:ncn*P][to_block]
            with np.errstate(invalid='ignore'):
                err = np.abs(got - want)/ent_scale[None, :]
            return np.where(np.isnan(err), np.inf, err), got, hsaco
        finally:
            h.close()

    # which wave writes jac[nd*P + e]: the one whose entry range holds the
    # FIRST entry of the 128-byte line (opty_device.h; the output of the
    # verification problem starts on a line)
    nd_, e_ = np.meshgrid(np.arange(ncn), np.arange(P), indexing='ij')
    first = ((nd_*P + e_)//16)*16 - nd_*P
    first = np.where(first < 0, first + P, first)   # (line began a node ago)
    if csr:
        first = e_          # a wave flushes its own rows (whole row chunks)

    lines = source.splitlines()
    err0, got0, _ = run(source)
    bad0 = err0 > 1e-6
    wrong0 = np.where(bad0.any(axis=0))[0]
    print('%s: %d lines; %d entries of %s wrong (e.g. %s)' % (
        tag, len(lines), len(wrong0), kernel, wrong0[:8]), flush=True)
    assert len(wrong0), 'this compiler builds the module correctly'

    # -- A: one kernel, one strip ---------------------------------------------
    ks = kernels(lines)
    i, j = ks[kernel]
    body = lines[i:j]
    cases = [k for k, ln in enumerate(body)
             if re.match(r'\s+case \d+: \{', ln)]
    tail = max(k for k, ln in enumerate(body) if 'default: break;' in ln)
    bounds = cases + [tail]
    # the strip whose wave wrote most of the wrong values
    strip, most = None, 0
    for c in range(len(cases)):
        m = [re.match(r'\s*// strip (\d+) (\d+)', ln)
             for ln in body[bounds[c]:bounds[c + 1]]]
        m = [x for x in m if x]
        if not m:
            continue
        a0, a1 = int(m[0].group(1)), int(m[-1].group(2))
        count = int((bad0 & (first >= a0) & (first < a1)).sum())
        if count > most:
            strip, most, e0, e1 = c, count, a0, a1
    assert strip is not None
    mine = (first >= e0) & (first < e1)     # what this strip's wave writes
    print('   strip %d = entries [%d, %d) wrote %d of the %d wrong values'
          % (strip, e0, e1, most, int(bad0.sum())), flush=True)

    def wrong_entries(err):
        return [int(e) for e in np.where((mine & (err > 1e-6)).any(axis=0))[0]]
    prologue = body[:cases[0]]
    # (drop the folded instance-constraint block at the kernel's top)
    if any('blockIdx.x >=' in ln for ln in prologue[:6]):
        a = next(k for k, ln in enumerate(prologue) if 'blockIdx.x >=' in ln)
        b = next(k for k in range(a, len(prologue))
                 if prologue[k].strip() == '}' and
                 prologue[k - 1].strip() == 'return;') + 1
        prologue = prologue[:a] + prologue[b:]
    case = body[bounds[strip]:bounds[strip + 1]]

    def module(case_lines):
        out = lines[:min(v[0] for v in ks.values())]
        for name, (a, b) in sorted(ks.items(), key=lambda kv: kv[1][0]):
            if name == kernel:
                out += prologue + case_lines + body[tail:]
            elif name == 'opty_uni':
                out += lines[a:b]
            else:
                out += shell(lines, a, b)
            out.append('')
        return '\n'.join(out) + '\n'

    text = module(case)
    err, got, hsaco = run(text)
    own = [int(e) for e in np.where(mine.any(axis=0))[0]]
    wrong = wrong_entries(err)
    print('A: strip %d = entries [%d, %d) alone: %d lines, %d entries still '
          'wrong' % (strip, e0, e1, len(text.splitlines()), len(wrong)),
          flush=True)
    best = (text, case, wrong, err, got, hsaco)
    multi = False
    parts = None            # strip -> its `case` lines (several strips kept)
    if not wrong:
        # One register allocation (and one `switch`) serves every `case` of
        # the kernel: alone the strip is compiled correctly.  Keep all
        # strips, then (A2) drop the other ones one at a time, largest
        # first, while the strip stays wrong, and (A3) replace the body of
        # every strip that has to stay by the smallest one's (a handful of
        # constant stores): what is left is the context the fault needs.
        print('   the fault needs its neighbours: every strip of the kernel '
              'kept, the other kernels as shells', flush=True)
        multi = True
        parts = {c: body[bounds[c]:bounds[c + 1]] for c in range(len(cases))}

        def assemble(parts):
            out = []
            for c in sorted(parts):
                out += parts[c]
            return out

        def judge(parts, label):
            nonlocal best
            t = module(assemble(parts))
            try:
                err, got, hsaco = run(t)
            except Exception as exc:        # noqa
                print('   %s: %s' % (label, str(exc)[:80]))
                return False
            w = wrong_entries(err)
            print('   %s: %d lines, %d entries of strip %d wrong' % (
                label, len(t.splitlines()), len(w), strip), flush=True)
            if w:
                best = (t, dict(parts), w, err, got, hsaco)
            return bool(w)

        assert judge(parts, 'A: all strips'), \
            'the fault needs the other kernels too: not reduced'
        for c in sorted((c for c in parts if c != strip),
                        key=lambda c: -len(parts[c])):
            trial = {k: v for k, v in parts.items() if k != c}
            if judge(trial, 'A2: without strip %d (%d lines)'
                     % (c, len(parts[c]))):
                parts = trial
        small = min((c for c in parts if c != strip),
                    key=lambda c: len(parts[c]))
        for c in sorted((c for c in parts if c not in (strip, small)),
                        key=lambda c: -len(parts[c])):
            if len(parts[c]) <= len(parts[small]) + 8:
                continue
            stand_in = [re.sub(r'case \d+:', 'case %d:' % c, parts[small][0])
                        ] + parts[small][1:]
            trial = dict(parts)
            trial[c] = stand_in
            if judge(trial, 'A3: strip %d (%d lines) replaced by strip %d\'s '
                     'body' % (c, len(parts[c]), small)):
                parts = trial
        print('   strips kept: %s' % sorted(parts), flush=True)
        # second sweep of A2 (what the stand-ins made unnecessary)
        for c in sorted((c for c in parts if c != strip),
                        key=lambda c: -len(parts[c])):
            trial = {k: v for k, v in parts.items() if k != c}
            if len(trial) > 1 and judge(
                    trial, 'A2\': without strip %d' % c):
                parts = trial
        print('   strips kept: %s' % sorted(parts), flush=True)

    # -- B: entries -------------------------------------------------------------
    R = 48 if meta['chunk'] == 32 else meta['chunk'] + 16
    TS = 65

    def blank(case_lines, entries):
        """Ring writes of the strip's ``entries`` store 0.0."""
        rows = {(e % R)*TS for e in entries}
        # the k-th write to a ring row belongs to entry e0 + position: walk
        # the writes in order and count per row
        out, v = [], e0
        for ln in case_lines:
            m = RING.match(ln)
            if m:
                if v in entries and int(m.group(2)) == (
                        (v - e0)*TS if csr else (v % R)*TS):
                    ln = m.group(1) + '0.0;'
                v += 1
            out.append(ln)
        return dce(out)

    def attempt(entries_to_blank, label):
        nonlocal best
        if multi:
            cand = dict(best[1])
            cand[strip] = blank(cand[strip], set(entries_to_blank))
            t = module(assemble(cand))
        else:
            cand = blank(best[1], set(entries_to_blank))
            t = module(cand)
        try:
            err, got, hsaco = run(t)
        except Exception as exc:            # noqa: a candidate may not compile
            print('   %s: %s' % (label, str(exc)[:80]))
            return False
        still = [e for e in wrong_entries(err)
                 if e not in entries_to_blank]
        print('   %s: %d lines, %d wrong entries left' % (
            label, len(t.splitlines()), len(still)), flush=True)
        if still:
            best = (t, cand, still, err, got, hsaco)
            return True
        return False

    right = [e for e in own if e not in best[2]]
    attempt(right, 'B: entries that come out right store 0.0')
    steps = 0
    while len(best[2]) > 1 and steps < 24:
        steps += 1
        size = len(best[0])
        half = best[2][len(best[2])//2:]
        if not attempt(half, 'B: dropping %d of the wrong entries'
                       % len(half)):
            half = best[2][:len(best[2])//2]
            if not attempt(half, 'B: dropping the other %d' % len(half)):
                break
        if len(best[0]) >= size:
            break           # (a fault that is not in the values: nothing
            #                 left to blank)

    text, case, wrong, err, got, hsaco = best
    out = os.path.join(REPO, 'gpurun_out', 'reduced_%s.hip' % tag)
    with open(out, 'w') as f:
        f.write(text)
    res = hb.kernel_resources(hsaco)[kernel]
    with open(out[:-4] + '.txt', 'w') as f:
        f.write('reduced from tools/o3_repro/%s.hip.xz (%d lines) to %d lines '
                'in %d compile-and-run steps; hipcc %s --offload-arch=gfx950 '
                '--genco\n' % (tag, len(lines), len(text.splitlines()),
                               tried[0], info['opt_level']))
        f.write('%s: %d VGPRs, %d spilled VGPRs, %d spilled SGPRs\n' % (
            kernel, res['.vgpr_count'], res['.vgpr_spill_count'],
            res['.sgpr_spill_count']))
        f.write('problem %s, N = %d nodes, free = default_rng(7).uniform(-1, '
                '1); jac[node*%d + entry], expected = the expression DAG '
                'evaluated one operation per instruction '
                '(opty_hip_tape_run)\n' % (info['problem'], N, P))
        for e in wrong[:12]:
            for nd in range(ncn):
                if mine[nd, e] and err[nd, e] > 1e-6:
                    f.write('entry %4d node %3d: expected % .15e  actual '
                            '% .15e\n' % (e, nd, want[nd, e], got[nd, e]))
    print(open(out[:-4] + '.txt').read())


if __name__ == '__main__':
    main()
