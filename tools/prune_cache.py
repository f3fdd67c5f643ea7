#!/usr/bin/env python
"""Developer tool (CPU): keeps in ``opty_amd/_cache`` only the code objects a
run asked for.

    OPTY_CACHE_MANIFEST=/tmp/used.txt python -c "import __graft_entry__ as g; g.build()"
    python tools/prune_cache.py /tmp/used.txt [more manifests ...]

(``hip_backend.compile_module`` appends the name of every code object it is
asked for to ``$OPTY_CACHE_MANIFEST``.)  Everything else -- the candidates of
tuning runs, the builds of abandoned printer versions -- is deleted together
with its side files; the cache is a cache: anything missing is rebuilt on
demand."""
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
CACHE = os.path.join(REPO, 'opty_amd', '_cache')


def main():
    keep = set()
    for path in sys.argv[1:]:
        with open(path) as f:
            keep |= {ln.strip()[:-len('.hsaco')] for ln in f if ln.strip()}
    if not keep:
        sys.exit('no manifest given / empty manifest: nothing pruned')
    freed = kept = 0
    for name in os.listdir(CACHE):
        stem = name.split('.')[0]
        if not name.startswith('opty_') or stem in keep:
            kept += 1
            continue
        path = os.path.join(CACHE, name)
        freed += os.path.getsize(path)
        os.remove(path)
    print('%d files kept, %.0f MB freed' % (kept, freed/1e6))


if __name__ == '__main__':
    main()
