"""The C ABI from plain C (``tests/c_client/abi_client.c``, built with gcc
against ``include/opty_hip.h`` and ``libopty_hip.so``): no Python, no torch
and no C++ on the calling side -- what INTEGRATION.md promises a maintainer of
the reference.  The CPU half builds the client; the GPU half runs it on
problems with known parameters, instance constraints and a free node time
interval and compares its outputs with the Python host side's, bit for bit."""
import ctypes
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from examples import problems

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
SOURCE = os.path.join(REPO, 'tests', 'c_client', 'abi_client.c')


def _build(tmp_path):
    from opty_amd import hip_backend as hb
    hb.build_runtime_library()
    exe = str(tmp_path/'abi_client')
    cmd = ['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-O1',
           '-I', os.path.join(REPO, 'include'), SOURCE,
           '-L', os.path.dirname(hb.LIB_PATH), '-lopty_hip',
           '-Wl,-rpath,' + os.path.dirname(hb.LIB_PATH), '-o', exe]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    return exe


@pytest.mark.skipif(shutil.which('gcc') is None, reason='no gcc')
def test_c_client_builds_against_the_header(tmp_path):
    """The header is plain C99 (``-std=c99 -Wall -Wextra -Werror``) and every
    entry point the client uses links."""
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    # without a case file the client fails cleanly (no device is touched)
    proc = subprocess.run([exe], capture_output=True, text=True)
    assert proc.returncode == 2 and 'usage' in proc.stderr


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['host', 'shard'])
@pytest.mark.parametrize('name', ['config2_pendulum_small',
                                  'pend2_link_vardur_unkmass_small',
                                  'chaplygin_be_small'])
def test_c_client_matches_the_python_host(name, mode, tmp_path):
    """``mode='shard'``: the same outputs through the node-sharded calls of
    the ABI from plain C -- ``opty_hip_comm_create`` (the library's own RCCL
    communicator, a world of one), ``opty_hip_bcast_free``, two
    ``opty_hip_eval_shard`` launches into shard buffers, ``opty_hip_gather_v``
    and ``opty_hip_eval_instance``."""
    import opty_amd
    from opty_amd import hip_backend as hb
    exe = _build(tmp_path)
    col = opty_amd.ConstraintCollocator(**problems.build(name))
    assert col.num_known_input_trajectories == 0
    hip = col.hip
    free = problems.make_free(col.num_free, seed=4,
                              variable_duration=col._variable_duration)
    con = col.generate_constraint_function()(free)
    jac = np.array(col.generate_jacobian_function()(free))
    rows, cols = col.jacobian_indices()
    hsaco, meta = col._build_code_object()
    desc = hb._Desc(**col._descriptor(meta))
    par = np.array([float(col.known_parameter_map[p])
                    for p in col.known_parameters], dtype=np.float64)
    idx = col.instance_constraints_free_index_map \
        if col.num_instance_constraints else {}
    atoms = np.array([idx[f] for f in col._inst_atoms], dtype=np.int64)
    case = tmp_path/'case.bin'
    with open(case, 'wb') as f:
        f.write(bytes(desc))
        path = hsaco.encode()
        f.write(struct.pack('<q', len(path)))
        f.write(path)
        f.write(par.tobytes())
        f.write(struct.pack('<d', 0.0 if col._variable_duration
                            else float(col.node_time_interval)))
        f.write(atoms.tobytes())
        f.write(col._inst_rows.astype(np.int64).tobytes())
        f.write(col._inst_cols.astype(np.int64).tobytes())
        f.write(struct.pack('<q', col.num_free))
        f.write(np.ascontiguousarray(free, dtype=np.float64).tobytes())
    out = tmp_path/'out.bin'
    # one node: RCCL's bootstrap socket on the loopback interface (a GPU box
    # of this pool once sat 300 s in the client's shard mode -- the only part
    # of it that opens a socket -- with the bootstrap on the container's veth)
    env = dict(os.environ)
    env.setdefault('NCCL_SOCKET_IFNAME', 'lo')
    cmd = [exe, str(case), str(out)] + (['shard'] if mode == 'shard' else [])
    try:
        proc = subprocess.run(cmd, capture_output=True, text=True,
                              timeout=120, env=env)
    except subprocess.TimeoutExpired as err:
        import warnings
        warnings.warn('C client (%s) hung for 120 s, retried once; its '
                      'output: %r %r' % (mode, err.stdout, err.stderr))
        proc = subprocess.run(cmd, capture_output=True, text=True,
                              timeout=240, env=dict(env, NCCL_DEBUG='INFO'))
    assert proc.returncode == 0, proc.stderr
    raw = out.read_bytes()
    ncon, nnz = struct.unpack_from('<qq', raw, 0)
    assert (ncon, nnz) == (col.num_constraints, hip.nnz)
    at = 16
    c = np.frombuffer(raw, dtype=np.float64, count=ncon, offset=at)
    at += 8*ncon
    j = np.frombuffer(raw, dtype=np.float64, count=nnz, offset=at)
    at += 8*nnz
    r = np.frombuffer(raw, dtype=np.int64, count=nnz, offset=at)
    at += 8*nnz
    k = np.frombuffer(raw, dtype=np.int64, count=nnz, offset=at)
    if mode == 'host':
        np.testing.assert_array_equal(c, con)
        np.testing.assert_array_equal(j, jac)
    else:
        # node windows through other kernels of the module (fused / separate)
        # agree to rounding, not to the bit
        for got, want in ((c, con), (j, jac)):
            np.testing.assert_allclose(got, want, rtol=1e-11,
                                       atol=1e-11*np.abs(want).max())
    np.testing.assert_array_equal(r, rows)
    np.testing.assert_array_equal(k, cols)
    assert ctypes.sizeof(desc) == len(bytes(desc))
