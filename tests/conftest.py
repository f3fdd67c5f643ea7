import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


# every RCCL bootstrap of this suite is within one box: loopback, not
# whichever interface the container happens to have (tests/test_c_client.py)
os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    """``-m gpu`` tests are the only ones allowed to touch a device."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter):
    """Worst per-entry errors the parity checks saw (relative to the entry,
    and in units of the entry's rounding-error bound), so that drift is
    visible even while everything passes; also written to
    ``gpurun_out/parity_stats.json`` on a GPU box."""
    try:
        import golden_util as gu
    except ImportError:
        return
    if not gu.STATS:
        return
    import json
    worst = sorted(gu.STATS.items(),
                   key=lambda kv: -kv[1]['worst_bound_units'])
    tr = terminalreporter
    tr.write_line('parity: worst per-entry errors (top 8 of %d labels)'
                  % len(worst))
    for label, st in worst[:8]:
        tr.write_line('  %-44s rel %.2e   %.2f bound units   (%d entries)'
                      % (label[:44], st['worst_rel'],
                         st['worst_bound_units'], st['entries']))
    # entries that met the bar only through the tolerance floor (they miss
    # 1e-10 of their OWN value: cancellation), per label
    floored = sorted(((k, v) for k, v in gu.STATS.items()
                      if v.get('entries_passed_by_floor')),
                     key=lambda kv: -kv[1]['entries_passed_by_floor'])
    tr.write_line('parity: %d of %d labels have entries that pass only by '
                  'the floor%s' % (len(floored), len(worst),
                                   ':' if floored else ''))
    for label, st in floored[:12]:
        tr.write_line('  %-44s %7d of %d entries, worst rel %.2e%s'
                      % (label[:44], st['entries_passed_by_floor'],
                         st['entries'], st['worst_rel_passed_by_floor'],
                         '' if st.get('floor_capped') else '  (floor NOT '
                         'capped by the row maximum)'))
    out = os.path.join(REPO, 'gpurun_out')
    try:
        import torch
        gpu = torch.cuda.is_available()
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'parity_stats.json' if gpu
                               else 'parity_stats_cpu.json'), 'w') as f:
            json.dump(gu.STATS, f, indent=1, sort_keys=True)
    except Exception:
        pass
