"""Isomorphic sub-models of a collocation program's expression DAG.

A musculoskeletal model repeats itself: the four De Groote musculotendon
actuators of the gallery's muscle-driven leg, the two mirror-image legs and
the four contact points of the seven-segment biped, the sixteen muscles of
gait2d.  In the hash-consed DAG (``ir.py``) such instances are sub-DAGs of the
SAME shape over different leaves (other state rows, other parameters).  This
module finds them -- a structural hash with the leaves abstracted -- and
prices what evaluating them side by side in the lanes of a wave (a wave holds
64/k nodes x k instances) could buy.

The price list is simple enough to state here, and it decides the matter
before any kernel is printed (``profiles/r06_isomorphic.txt`` has the numbers
per problem):

* per-LANE work (what sets the latency of a wave that holds a SIMD alone, i.e.
  what bounds a launch that under-fills the chip -- a 1/8 node shard): the
  block's work minus ``(k - 1) w`` for every group of ``k`` instances of
  weight ``w`` -- Amdahl over the part of the block that is NOT repeated;
* SIMD-time (``sum(wave durations)/1024``, what bounds a full-size launch of
  these blocks): a wave of 64/k nodes executes the non-repeated remainder for
  64/k nodes instead of 64, so the remainder's SIMD-time grows ``k`` times while
  the instances' stays what it was -- lane vectorisation never lowers it.

``instance_groups`` is also what a cooperative (several waves per node block,
exchange through LDS) geometry would start from: ``interface`` counts the
values an instance hands to the rest of the block.
"""

from . import ir

_COMMUTATIVE = (ir.ADD, ir.MUL, ir.MAX, ir.MIN)


def default_leaf(dag):
    """Leaves of the per-node code: constants, inputs, node-invariant
    sub-expressions (they come from the table ``opty_uni`` fills)."""
    return lambda i: dag.op[i] in (ir.CONST, ir.INPUT) or dag.uni[i]


def shape_hashes(dag, nodes, is_leaf=None):
    """``{node: hash}`` of the sub-DAG under every non-leaf node of ``nodes``
    (ascending ids = topological order) with the leaves abstracted: two nodes
    get the same hash when the trees under them have the same operations in
    the same places (operands of commutative operations in any order)."""
    is_leaf = is_leaf or default_leaf(dag)
    shape = {}

    def sh(i):
        return 'L' if is_leaf(i) else shape[i]

    for i in nodes:
        if is_leaf(i):
            continue
        op = dag.op[i]
        if op == ir.POWI:
            key = (op, sh(dag.args[i][0]), dag.args[i][1])
        elif op == ir.SELECT:
            key = (op, dag.args[i][0]) + tuple(sh(j)
                                               for j in dag.args[i][1:])
        else:
            hs = [sh(j) for j in dag.args[i]]
            if op in _COMMUTATIVE:
                hs.sort(key=lambda h: (0, 0) if h == 'L' else (1, h))
            key = (op,) + tuple(hs)
        shape[i] = hash(key)
    return shape


def cone(dag, root, is_leaf):
    """Non-leaf nodes under ``root`` (``root`` included)."""
    seen, stack = set(), [root]
    while stack:
        i = stack.pop()
        if i in seen or is_leaf(i):
            continue
        seen.add(i)
        stack.extend(dag.operands(i))
    return seen


def instance_groups(dag, roots, weight, is_leaf=None, min_weight=60,
                    max_shared=0.25):
    """Groups of isomorphic, (nearly) disjoint sub-DAGs among the nodes
    needed for ``roots``.

    ``weight(i)``: cost of node ``i`` (the printer's operation weights).
    A group is kept when every instance weighs at least ``min_weight`` and
    the instances share at most ``max_shared`` of an instance's weight
    (mirror-image legs share the trunk's kinematics; sub-DAGs that share most
    of their work are the SAME sub-model seen from two outputs, not two
    instances).  Groups are chosen greedily by the lane work they would save,
    ``(k - 1) w``, and never overlap.

    Returns a list of dicts: ``roots`` (one per instance), ``k``, ``weight``
    (one instance), ``shared`` (weight common to all instances), ``saved`` =
    ``(k - 1)(weight - shared)``, ``interface`` (values of one instance --
    its root and inner nodes -- that the rest of the block reads) and
    ``leaves`` (distinct leaves of one instance)."""
    is_leaf = is_leaf or default_leaf(dag)
    need = [i for i in dag.reachable(roots) if not is_leaf(i)]
    shape = shape_hashes(dag, need, is_leaf)
    by_shape = {}
    for i in need:
        by_shape.setdefault(shape[i], []).append(i)
    users = {}
    for i in need:
        for j in dag.operands(i):
            if not is_leaf(j):
                users.setdefault(j, []).append(i)
    outputs = set(roots)
    cands = []
    for members in by_shape.values():
        if len(members) < 2:
            continue
        c0 = cone(dag, members[0], is_leaf)
        w = sum(weight(v) for v in c0)
        if w < min_weight:
            continue
        cones = [c0] + [cone(dag, m, is_leaf) for m in members[1:]]
        # instances must be distinct sub-DAGs: drop members whose cone is
        # (nearly) another member's
        keep, kept = [], []
        for m, c in zip(members, cones):
            if all(sum(weight(v) for v in (c & o)) <= max_shared*w
                   for o in kept):
                keep.append(m)
                kept.append(c)
        if len(keep) < 2:
            continue
        common = set.intersection(*kept)
        shared = sum(weight(v) for v in common)
        cands.append(((len(keep) - 1)*(w - shared), w, shared, keep, kept))
    cands.sort(key=lambda t: -t[0])
    taken, out = set(), []
    for saved, w, shared, keep, kept in cands:
        own = [c - set.intersection(*kept) for c in kept]
        if any(c & taken for c in own):
            continue
        for c in own:
            taken |= c
        c0 = kept[0]
        interface = sum(1 for v in c0 if v in outputs or any(
            u not in c0 for u in users.get(v, ())))
        leaves = set()
        for v in c0:
            leaves.update(j for j in dag.operands(v) if is_leaf(j) and
                          dag.op[j] != ir.CONST)
        out.append(dict(roots=list(keep), k=len(keep), weight=w,
                        shared=shared, saved=saved, interface=interface,
                        leaves=len(leaves)))
    return out


def lane_vectorisation_bounds(total, groups):
    """What side-by-side evaluation of ``groups`` in the lanes of a wave can
    reach for a block of ``total`` weighted operations per node:

    ``lane_work``   per-lane operations (the wave's latency) after every
                    group's ``(k - 1)`` repeated instances have moved to other
                    lanes, plus two LDS accesses per interface value and
                    instance for the exchange;
    ``latency_gain`` ``total / lane_work``;
    ``k``            lanes per node (the largest group's: one wave geometry);
    ``simd_time``    relative SIMD-time of the launch, ``(k (total - repeated)
                    + repeated) / total``: the non-repeated remainder is
                    executed for 64/k nodes per wave instead of 64."""
    if not groups:
        return dict(lane_work=total, latency_gain=1.0, k=1, simd_time=1.0,
                    repeated=0)
    k = max(g['k'] for g in groups)
    saved = sum(g['saved'] for g in groups)
    exchange = sum(2*g['interface']*g['k'] for g in groups)
    repeated = sum(g['k']*(g['weight'] - g['shared']) for g in groups)
    lane = total - saved + exchange
    return dict(lane_work=lane, latency_gain=total/float(lane), k=k,
                simd_time=(k*(total - repeated) + repeated)/float(total),
                repeated=repeated)
