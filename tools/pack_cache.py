#!/usr/bin/env python
"""Developer tool (CPU): re-bundles the code objects of ``opty_amd/_cache``
as COMPRESSED offload bundles (zstd; what ``hipcc --offload-compress`` writes
and ``hipModuleLoad`` / ``clang-offload-bundler`` read), in place.  New builds
are compressed by ``hip_backend.compile_module`` already; this converts the
ones that were built before.

    python tools/pack_cache.py [cache dir]
"""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
BUNDLER = '/opt/rocm/lib/llvm/bin/clang-offload-bundler'
TARGETS = '--targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950'
LEVEL = os.environ.get('OPTY_PACK_LEVEL', '19')


def pack(path):
    with open(path, 'rb') as f:
        if f.read(4) == b'CCOB':
            return 0
    before = os.path.getsize(path)
    with tempfile.TemporaryDirectory() as tmp:
        host, dev, out = (os.path.join(tmp, n) for n in ('h.o', 'd.o', 'o'))
        subprocess.run([BUNDLER, '--unbundle', '--type=o', '--input=' + path,
                        TARGETS, '--output=' + host, '--output=' + dev],
                       check=True, capture_output=True)
        subprocess.run([BUNDLER, '--type=o', TARGETS, '--input=' + host,
                        '--input=' + dev, '--output=' + out, '-compress',
                        '-compression-level=' + LEVEL],
                       check=True, capture_output=True)
        st = os.stat(path)
        os.replace(out, path)
        os.utime(path, (st.st_atime, st.st_mtime))   # side files stay valid
    return before - os.path.getsize(path)


def main():
    cache = sys.argv[1] if len(sys.argv) > 1 else \
        os.path.join(REPO, 'opty_amd', '_cache')
    files = [os.path.join(cache, n) for n in sorted(os.listdir(cache))
             if n.endswith('.hsaco')]
    with ThreadPoolExecutor(os.cpu_count() or 4) as ex:
        saved = list(ex.map(pack, files))
    print('%d code objects, %d re-bundled, %.0f MB saved'
          % (len(files), sum(s > 0 for s in saved), sum(saved)/1e6))


if __name__ == '__main__':
    main()
