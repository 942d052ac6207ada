"""ctypes loader of the plain-C oracle (oracle/star_oracle_c.c -> oracle/_build/libstar_oracle_c.so).
TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by star_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libstar_oracle_c.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _lib = C.CDLL(_SO)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def late_fusion(maps, trans, num_agent, batch, ego_maps=None):
    """Same contract as star_oracle.late_fusion (multiagent_mae.py:870-944), computed by the C restatement."""
    x = np.ascontiguousarray(maps, dtype=np.float32)
    ab, c, h, w = x.shape
    a = ab // batch
    e = None if ego_maps is None else np.ascontiguousarray(ego_maps, dtype=np.float32)
    t = np.ascontiguousarray(trans)
    assert t.dtype in (np.float64, np.float32) and t.shape == (batch, a, a, 4, 4)
    na = np.ascontiguousarray(np.asarray(num_agent).reshape(batch, -1), dtype=np.int64)
    assert na.shape[1] == a
    out = np.empty_like(x)
    rc = lib().oc_late_fusion(_p(x), None if e is None else _p(e), _p(t), int(t.dtype == np.float64), _p(na), _p(out), batch, a, c, h, w)
    assert rc == 0
    return out


def shuffle_ids(noise):
    """Stable argsort and its inverse (multiagent_mae.py:811-814)."""
    n = np.ascontiguousarray(noise, dtype=np.float32)
    r, l = n.shape
    s, inv = np.empty((r, l), np.int64), np.empty((r, l), np.int64)
    assert lib().oc_argsort_pair(_p(n), _p(s), _p(inv), r, l) == 0
    return s, inv


def confusion(pred, target):
    p = np.ascontiguousarray(pred, dtype=np.float32)
    t = np.ascontiguousarray(target, dtype=np.float32)
    conf = np.zeros(4, np.int64)
    assert lib().oc_confusion(_p(p), _p(t), C.c_int64(p.size), _p(conf)) == 0
    return conf.reshape(2, 2)


def voxel_dense(frames, dims=(256, 256, 13)):
    """list of [n,3] index arrays -> float32 [F, Y, X, Z] (MultiTempDet.py:134-143 per frame); IndexError like numpy."""
    counts = [len(f) for f in frames]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(f, dtype=np.int64).reshape(-1, 3) for f in frames] + [np.zeros((0, 3), np.int64)], 0))
    x, y, z = dims
    out = np.empty((len(frames), y, x, z), np.float32)
    rc = lib().oc_voxel_dense(_p(flat), _p(off), _p(out), len(frames), x, y, z)
    if rc == -2:
        raise IndexError("voxel index out of bounds")
    assert rc == 0
    return out
