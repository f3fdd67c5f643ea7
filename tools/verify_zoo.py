"""Developer tool (GPU): holds every prebuilt code object of the zoo
(``__graft_entry__.prebuilt_collocators``) to the instruction tape
(``ConstraintCollocator._verify_build``), lets refused builds be replaced
(``_verified_alternative``: recorded as "pinned" entries of the plan file) and
writes the verdicts + the updated plan file under ``gpurun_out/``.

Usage: verify_zoo.py [--hot] [name substring ...]
  --hot: only the builds at the register limit (default: every build)"""
import json
import logging
import os
import shutil
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
args = [a for a in sys.argv[1:] if not a.startswith('--')]
os.environ['OPTY_CROSS_CHECK'] = 'hot' if '--hot' in sys.argv else 'all'
logging.basicConfig(level=logging.WARNING)
import __graft_entry__ as ge
from opty_amd import hip_backend as hb, launch_plan

out = []
t00 = time.time()
for name, kw, col in ge.prebuilt_collocators():
    label = '%s %s' % (name, kw or '')
    if args and not any(a in label for a in args):
        continue
    t0 = time.time()
    try:
        hsaco, meta = col._build_code_object()
        hot = hb.high_pressure_kernels(hsaco)
        if not hot and os.environ['OPTY_CROSS_CHECK'] == 'hot':
            continue
        col.hip
        v = col._build_verdict or {}
        rec = dict(problem=label, ok=v.get('ok'), worst=v.get('worst'),
                   errors=v.get('errors'), kernels=v.get('kernels'),
                   replacement=v.get('replacement'),
                   refused=v.get('refused'), seconds=time.time() - t0)
        col.hip.close()
    except Exception as exc:        # noqa: keep going, report
        rec = dict(problem=label, ok=False,
                   error='%s: %s' % (type(exc).__name__, str(exc)[:800]))
    out.append(rec)
    print('%-70s %s worst %s%s' % (
        label[:70], 'ok ' if rec.get('ok') else 'FAILED',
        '%.2e' % rec['worst'] if rec.get('worst') is not None else '-',
        '  REPLACED by %s' % rec['replacement'] if rec.get('replacement')
        else ''), flush=True)
    if rec.get('error'):
        print('   ', rec['error'], flush=True)
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
with open(os.path.join(REPO, 'gpurun_out', 'verify_zoo.json'), 'w') as f:
    json.dump(out, f, indent=1)
shutil.copy(launch_plan.plan_path() or launch_plan.DEFAULT_FILE,
            os.path.join(REPO, 'gpurun_out', 'launch_plans.json'))
print('%d builds checked, %d failed, %d replaced, %.0f s' % (
    len(out), sum(not r.get('ok') for r in out),
    sum(bool(r.get('replacement')) for r in out), time.time() - t00))
