"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes front-end of oracle/voxel_pooling_ref.c
(reference: ops/voxel_pooling/src/voxel_pooling_forward_cuda.cu:9-36, ops/voxel_pooling/voxel_pooling.py:10-69)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libsgv3d_oracle.so")
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        i, p = ctypes.c_int, ctypes.c_void_p
        L.sgv3d_oracle_voxel_pooling_forward.argtypes = [i] * 6 + [p] * 4
        L.sgv3d_oracle_voxel_pooling_forward.restype = None
        L.sgv3d_oracle_voxel_pooling_forward_omp.argtypes = [i] * 6 + [p] * 4 + [i]
        L.sgv3d_oracle_voxel_pooling_forward_omp.restype = None
        L.sgv3d_oracle_voxel_pooling_backward.argtypes = [i] * 5 + [p] * 3
        L.sgv3d_oracle_voxel_pooling_backward.restype = None
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def forward(geom_xyz, feats, voxel_num, threads=1, want_pos_memo=True):
    """geom_xyz int32 [B,...,3], feats f32 [B,...,C], voxel_num (X,Y,Z)
    -> out f32 [B,C,Y,X] (NCHW like the reference returns), pos_memo int32 [B,N,3]."""
    B = geom_xyz.shape[0]
    C = feats.shape[-1]
    g = np.ascontiguousarray(geom_xyz, np.int32).reshape(B, -1, 3)
    f = np.ascontiguousarray(feats, np.float32).reshape(B, -1, C)
    N = g.shape[1]
    assert f.shape[1] == N
    X, Y, Z = (int(v) for v in voxel_num)
    out = np.zeros((B, Y, X, C), np.float32)
    pm = np.full((B, N, 3), -1, np.int32) if want_pos_memo else None
    pmp = _p(pm) if pm is not None else None
    if threads > 1:
        lib().sgv3d_oracle_voxel_pooling_forward_omp(B, N, C, X, Y, Z, _p(g), _p(f), _p(out), pmp, threads)
    else:
        lib().sgv3d_oracle_voxel_pooling_forward(B, N, C, X, Y, Z, _p(g), _p(f), _p(out), pmp)
    return np.ascontiguousarray(out.transpose(0, 3, 1, 2)), pm


def forward_nhwc_inplace(g, f, out, X, Y, Z, threads=1):
    """Timing entry: no allocation, NHWC out accumulated in place (cpu_baseline)."""
    B, N, C = f.shape
    if threads > 1:
        lib().sgv3d_oracle_voxel_pooling_forward_omp(B, N, C, X, Y, Z, _p(g), _p(f), _p(out), None, threads)
    else:
        lib().sgv3d_oracle_voxel_pooling_forward(B, N, C, X, Y, Z, _p(g), _p(f), _p(out), None)


def backward(pos_memo, grad_out, num_channels):
    """pos_memo int32 [B,N,3], grad_out f32 [B,C,Y,X] -> grad_feats f32 [B,N,C]."""
    pm = np.ascontiguousarray(pos_memo, np.int32)
    go = np.ascontiguousarray(grad_out, np.float32)
    B, N, _ = pm.shape
    _, C, Y, X = go.shape
    assert C == num_channels
    gi = np.empty((B, N, C), np.float32)
    lib().sgv3d_oracle_voxel_pooling_backward(B, N, C, X, Y, _p(pm), _p(go), _p(gi))
    return gi
