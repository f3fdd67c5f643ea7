#!/usr/bin/env python
"""Developer tool (CPU): static checks on the ISA of a generated kernel --
does every spill reload follow its spill on EVERY path?

    llvm-objdump -d <unbundled code object> > k.s
    python tools/isa_spill_check.py k.s opty_jac

Builds the kernel's control-flow graph from the disassembly (direct branches
only: the generated kernels have no indirect ones) and runs a forward
"must be defined" analysis for (i) the lanes of the VGPRs that carry spilled
SGPRs (v_writelane / v_readlane), (ii) AGPR copies (v_accvgpr_write / _read /
_mov), (iii) scratch slots (scratch_store / scratch_load offsets) and
(iv) whole vector registers (first operand = destination; partial writes
under a narrowed exec mask count as definitions).  r05: the frozen wrong build
tools/o3_repro/one_legged_park_spill_O2 -- whose output follows a register
poison pattern (profiles/r05_poison_probe.txt) -- passes all four, in its
wrong opty_jac and in its right opty_conjac alike, and every scratch load is
covered by a vmcnt wait before its first use: whatever it reads uninitialised,
it is not a spill slot that some path skips.  (Check (iv) reports false
positives where a value is defined in two blocks guarded by complementary exec
masks -- the two argument-range branches of an inlined sincos -- and each is
skipped by its own s_cbranch_execz: the frozen biped module shows 30 such
registers in its wrong kernels; long branches -- s_getpc / s_add / s_setpc --
are resolved.)"""
import collections
import re
import sys

_HEAD = r'''txt=open(FILE).read()
kernel=KERNEL
i=txt.index('<%s>:'%kernel)
rest=txt[i:].split('\n\n')[0] if False else txt[i:]
# end at next symbol label
m=re.search(r'\n[0-9a-f]+ <\w+>:', rest[10:])
body=rest[:m.start()+10] if m else rest
ins=[]
for l in body.splitlines()[1:]:
    m=re.match(r'\s*(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', l)
    if m: ins.append((int(m.group(3),16), m.group(1), m.group(2)))
addr_index={a:k for k,(a,_,_) in enumerate(ins)}
n=len(ins)
succ=[[] for _ in range(n)]
for k,(a,op,args) in enumerate(ins):
    nxt=ins[k+1][0] if k+1<n else None
    if op=='s_endpgm': continue
    if op=='s_branch' or op.startswith('s_cbranch'):
        off=int(args.split()[-1]); 
        if off>=32768: off-=65536
        tgt=a+4+off*4
        assert tgt in addr_index, (hex(a),op,args,hex(tgt))
        succ[k].append(addr_index[tgt])
        if op!='s_branch' and nxt is not None: succ[k].append(k+1)
    elif 'setpc' in op or 'swappc' in op:
        # a long branch: s_getpc_b64 s[a:b]; s_add_u32 sa, sa, LIT;
        # s_addc_u32 sb, sb, 0; s_setpc_b64 s[a:b]
        tgt=None
        if k>=3 and ins[k-3][1]=='s_getpc_b64' and ins[k-2][1]=='s_add_u32':
            lit=int(ins[k-2][2].split(',')[-1].strip(), 0)
            if lit >= 1 << 31: lit -= 1 << 32
            tgt=ins[k-2][0]+lit
        if tgt in addr_index:
            succ[k].append(addr_index[tgt])
        else:
            print('unresolved indirect branch at',hex(a))
    else:
        if nxt is not None: succ[k].append(k+1)
pred=[[] for _ in range(n)]
for k in range(n):
    for t in succ[k]: pred[t].append(k)
# must-written lanes: forward dataflow at instruction granularity (n ~ 1e4: fine with worklist on block leaders)
leaders={0}
for k in range(n):
    if len(succ[k])!=1 or succ[k][0]!=k+1:
        for t in succ[k]: leaders.add(t)
        if k+1<n: leaders.add(k+1)
leaders=sorted(leaders)
blk_of={}
blocks=[]
for bi,s in enumerate(leaders):
    e=leaders[bi+1] if bi+1<len(leaders) else n
    blocks.append((s,e))
    for k in range(s,e): blk_of[k]=bi
'''

def regs(tok, kind='v'):
    out=set()
    for m in re.finditer(r'(?<![a-z0-9_])%s\[(\d+):(\d+)\]'%kind, tok):
        out|={'%s%d'%(kind,q) for q in range(int(m.group(1)), int(m.group(2))+1)}
    for m in re.finditer(r'(?<![a-z0-9_\[:])%s(\d+)(?![\d:\]])'%kind, tok):
        out.add('%s%s'%(kind,m.group(1)))
    return out
NO_DEST=('v_cmp','v_cmpx','buffer_store','global_store','scratch_store','ds_write','flat_store','s_','v_readlane','v_readfirstlane','v_nop')
def analyse(FILE, KERNEL, show=15):
    g = {'FILE': FILE, 'KERNEL': KERNEL, 're': re, 'collections': collections}
    exec(_HEAD, g)
    ins, blocks, succ, blk_of, n = g['ins'], g['blocks'], g['succ'], g['blk_of'], g['n']
    bsucc=[set() for _ in blocks]; bpred=[set() for _ in blocks]
    for bi,(s,e) in enumerate(blocks):
        for t in succ[e-1]:
            bsucc[bi].add(blk_of[t]); bpred[blk_of[t]].add(bi)
    def du(k):
        a,op,args=ins[k]
        p=[x.strip() for x in re.split(r',\s*(?![^\[]*\])', args)]
        d=set(); u=set()
        if op.startswith(NO_DEST) or not p or not p[0]:
            for t in p: u|=regs(t)|regs(t,'a')
            if op in ('v_readlane_b32','v_readfirstlane_b32'): u=regs(','.join(p[1:]))
            return d,u
        if op=='v_writelane_b32':
            return regs(p[0]), set()        # partial write: counts as a definition
        d=regs(p[0])|regs(p[0],'a')
        # v_mad/div_scale with sgpr carry-out: second operand may be SGPR dest; ignore
        for t in p[1:]: u|=regs(t)|regs(t,'a')
        if 'fmac' in op or op.startswith(('v_mac','v_dot2c','v_mfma')): u|=regs(p[0])
        return d,u
    gen=[set() for _ in blocks]
    for bi,(s,e) in enumerate(blocks):
        for k in range(s,e): gen[bi]|=du(k)[0]
    entry={'v0'}
    IN=[None]*len(blocks); OUT=[None]*len(blocks)
    work=collections.deque([0])
    while work:
        b=work.popleft()
        ps=[OUT[p] for p in bpred[b] if OUT[p] is not None]
        if b==0: inn=set(entry)
        elif ps: inn=set.intersection(*ps)
        else: continue
        out=inn|gen[b]
        if IN[b]!=inn or OUT[b]!=out:
            IN[b]=inn; OUT[b]=out
            for t in bsucc[b]: work.append(t)
    bad=collections.Counter(); shown=0
    for bi,(s,e) in enumerate(blocks):
        if IN[bi] is None: continue
        cur=set(IN[bi])
        for k in range(s,e):
            d,u=du(k)
            for r in sorted(u):
                if r not in cur:
                    bad[r]+=1
                    if shown<show: print('   read of %s before any write on some path: %s %s %s'%(r, hex(ins[k][0]), ins[k][1], ins[k][2])); shown+=1
            cur|=d
    print(KERNEL, 'blocks', len(blocks), 'registers read before written on some path:', len(bad), sorted(bad)[:40])


def lanes(FILE, KERNEL):
    """(i)-(iii): spill lanes, AGPR copies, scratch slots."""
    g = {'FILE': FILE, 'KERNEL': KERNEL, 're': re, 'collections': collections}
    exec(_HEAD, g)
    ins, blocks, succ, blk_of = g['ins'], g['blocks'], g['succ'], g['blk_of']
    bsucc = [set() for _ in blocks]
    bpred = [set() for _ in blocks]
    for bi, (s, e) in enumerate(blocks):
        for t in succ[e - 1]:
            bsucc[bi].add(blk_of[t])
            bpred[blk_of[t]].add(bi)
    W = {'dword': 1, 'dwordx2': 2, 'dwordx3': 3, 'dwordx4': 4}

    def du(k):
        a, op, args = ins[k]
        p = [x.strip() for x in args.split(',')]
        d, u = set(), set()
        if op == 'v_writelane_b32':
            d.add(('lane', p[0], p[2]))
        elif op == 'v_readlane_b32':
            u.add(('lane', p[1], p[2]))
        elif op == 'v_accvgpr_write_b32':
            d.add(p[0])
        elif op == 'v_accvgpr_read_b32':
            u.add(p[1])
        elif op == 'v_accvgpr_mov_b32':
            d.add(p[0])
            u.add(p[1])
        elif op.startswith(('scratch_store', 'scratch_load')):
            m = re.search(r'offset:(\d+)', args)
            off = int(m.group(1)) if m else 0
            slots = {'scr%d' % (off + 4*q)
                     for q in range(W[op.split('_')[-1]])}
            (d if 'store' in op else u).update(slots)
        return d, u
    gen = [set() for _ in blocks]
    for bi, (s, e) in enumerate(blocks):
        for k in range(s, e):
            gen[bi] |= du(k)[0]
    IN, OUT = [None]*len(blocks), [None]*len(blocks)
    work = collections.deque([0])
    while work:
        b = work.popleft()
        ps = [OUT[q] for q in bpred[b] if OUT[q] is not None]
        if b == 0:
            inn = set()
        elif ps:
            inn = set.intersection(*ps)
        else:
            continue
        out = inn | gen[b]
        if IN[b] != inn or OUT[b] != out:
            IN[b], OUT[b] = inn, out
            work.extend(bsucc[b])
    bad = 0
    for bi, (s, e) in enumerate(blocks):
        if IN[bi] is None:
            continue
        cur = set(IN[bi])
        for k in range(s, e):
            d, u = du(k)
            for r in u:
                if r not in cur:
                    bad += 1
                    print('   reload before its spill on some path: %s %s %s'
                          % (hex(ins[k][0]), ins[k][1], ins[k][2]))
            cur |= d
    print('%s: %d blocks, %d spill reloads not dominated by their spill'
          % (KERNEL, len(blocks), bad))


if __name__ == '__main__':
    lanes(sys.argv[1], sys.argv[2])
    analyse(sys.argv[1], sys.argv[2])
