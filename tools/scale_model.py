#!/usr/bin/env python
"""Reads the lines of ``tools/scale_rehearsal.sh`` (bench.py with 1, 2, 4 and
8 ranks, the multi-rank ones oversubscribed on one GPU), REQUIRES every one of
them to have certified itself against the reference's golden record through
every re-assembly variant, and prints the predicted 1 -> 8 curve as JSON:

* ``compute_only``: the shard launches measured on this GPU (``config3_shards``
  of the 1-GPU line: each rank of an N-GPU node runs exactly that launch, no
  data-path collective) -- what ``bench.py --gpus N`` reports as ``value``;
* ``gather``: + the point-to-point gather-v to one GPU over xGMI: every peer
  sends its Jacobian slice and constraint block over its OWN link to the
  root (7 links x ~153 GB/s per direction, MI355X_MICROARCH.md; a link carries
  one peer's (810.4 + 17.6)/N MB);
* ``to_host``: + every rank's shard over its own PCIe link into the shared
  page-locked vector, varying entries only after the first call (the link
  rate measured on this box by ``host_path_ms.pcie``), the host scatter
  overlapped as on one GPU.

The model is the one of DESIGN.md section 7; the point of writing it down
BEFORE a node exists is that the first real SCALE_r0N.json gets compared with
a prediction instead of being explained afterwards."""
import json
import os
import sys

XGMI_LINK_GBPS = 153.0          # per direction and link (microarch guide)


def last_json(path):
    with open(path) as f:
        lines = [ln for ln in f if ln.startswith('{')]
    return json.loads(lines[-1])


def main(out):
    one = last_json(os.path.join(out, 'n1.json'))
    cfg = one['config']
    assert cfg['verify']['ok'] is True, cfg['verify']
    whole_ms = cfg['kernel_ms']['opty_conjac']
    shards = cfg['other_configs']['config3_shards']
    host = cfg['host_path_ms']
    link = host['pcie']['link_GBps']['d2h']
    M, P, ncn = 22, 990, 99999
    jac_mb, con_mb = 8e-6*P*ncn, 8e-6*M*ncn
    moved = host['moved_entries_per_block']
    rehearsed = {}
    for n in (2, 4, 8):
        line = last_json(os.path.join(out, 'n%d.json' % n))
        ver = line['config']['verify']
        assert ver['ok'] is True and ver['ranks'] == n, (n, ver)
        labels = ' '.join(ver['checked'])
        variants = line['config']['variants']
        for tag in ('benched launch', 'gather', 'gather_c_abi', 'to_host',
                    'callbacks'):
            assert tag in labels, (n, tag, ver['checked'])
        assert 'error' not in variants['gather_c_abi'], variants
        nt = last_json(os.path.join(out, 'n%d_notorch_r0.json' % n))
        assert nt['config']['verify']['ok'] is True and \
            nt['config']['verify']['ranks'] == n, nt['config']['verify']
        assert 'torch imported: False' in nt['config']['host']
        rehearsed[n] = {
            'verify_worst_rel': ver['worst_rel'],
            'variants_checked': sorted(variants),
            'no_torch_verify_worst_rel': nt['config']['verify']['worst_rel'],
            'oversubscribed': True}
    curve = {1: {'compute_only_ms': whole_ms, 'gather_ms': whole_ms,
                 'to_host_ms': host['jac'] + host['con']}}
    for n in (2, 4, 8):
        t = shards['shard_1of%d' % n]['fused_ms']
        per_peer_mb = (jac_mb + con_mb)/n
        gather = t + per_peer_mb/(XGMI_LINK_GBPS*1e3)*1e3       # ms
        down = 8e-6*moved*ncn/n/link + con_mb/n/link            # ms
        curve[n] = {'compute_only_ms': t, 'gather_ms': gather,
                    'to_host_ms': t + down + 0.9}   # + the 1-GPU path's fixed
        #                       part (first window, last chunk's scatter)
    base = curve[1]
    for n, c in curve.items():
        c['speedup'] = {k.replace('_ms', ''): base[k]/c[k]
                        for k in ('compute_only_ms', 'gather_ms',
                                  'to_host_ms')}
    json.dump({
        'what': 'predicted strong scaling of BASELINE config 4 (10-link, N = '
                '100 000) from launches measured on ONE MI355X + link model; '
                'no 8-GPU node was available to this build',
        'measured_on_this_gpu': {
            'whole_problem_ms': whole_ms,
            'shard_ms': {n: shards['shard_1of%d' % n]['fused_ms']
                         for n in (2, 4, 8)},
            'pcie_d2h_GBps': link, 'host_path_jac_ms': host['jac'],
            'moved_entries_per_block': moved},
        'assumed': {'xgmi_GBps_per_link_and_direction': XGMI_LINK_GBPS,
                    'host_path_fixed_ms': 0.9},
        'curve': curve,
        'rehearsed_with_n_ranks_on_one_gpu': rehearsed,
        'target': 'north_star: >= 6x at 8 GPUs (compute only: outputs left '
                  'distributed, as bench.py --gpus N reports `value`)',
    }, sys.stdout, indent=1)
    print()


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/rehearsal')
