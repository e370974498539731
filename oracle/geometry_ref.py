"""ORACLE — TEST INFRASTRUCTURE ONLY.

Never imported by the product path (``sgv3d_amd/``); only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may use it.

numpy (float32, explicit evaluation order) restatement of the reference's camera->BEV geometry:

* ``create_frustum``      follows /root/reference/layers/backbones/lss_fpn.py:325-348
* ``calib_prep``          follows lss_fpn.py:361 (sensor2virtual @ K^-1), :367 (sensor2ego @
                          sensor2virtual^-1), :390 (ida^-1); 4x4 products are sequential fp32
                          (that is what torch's small-bmm CPU kernel does: verified bit-equal in
                          tests/golden/make_golden.py); the 4x4 *inverse* is this build's own fixed
                          algorithm (``inv4``) because the reference delegates it to the platform's
                          LAPACK/MAGMA whose rounding is not reproducible across devices.
* ``geometry_points``     follows lss_fpn.py:350-370 (height2localtion) and :372-401 (get_geometry)
* ``quantise``            follows lss_fpn.py:487-488 with the *GPU* float->int32 semantics the
                          reference actually runs with (``.int()`` on a CUDA tensor: truncate toward
                          zero, saturate, NaN -> 0).

Pinned by tests/golden/geometry_*.npz, which hold outputs of the reference's own
``LSSFPN.create_frustum`` / ``get_geometry`` executed in the build container.
"""
import numpy as np

f32 = np.float32


# --------------------------------------------------------------------------------------------
# G1  create_frustum  (lss_fpn.py:325-348)
# --------------------------------------------------------------------------------------------
def _linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps, dtype=float32) on CPU: symmetric fill,
    step = (end-start)/(steps-1) in float32, first half fma(step, i, start), second half
    fma(-step, steps-1-i, end).  The fused multiply-add (one rounding) is what torch's vectorised
    kernel does (verified against the reference-generated fixtures, tests/golden/frustum.npz); it is
    emulated exactly here: a float32 x float32 product is exact in float64, and the float64 sum is
    then rounded once to float32 (operands are < 2^12, so no double-rounding case arises)."""
    out = np.empty(steps, f32)
    if steps == 1:
        out[0] = f32(start)
        return out
    start = f32(start)
    end = f32(end)
    step = f32(f32(end - start) / f32(steps - 1))
    half = steps // 2
    for i in range(steps):
        if i < half:
            out[i] = f32(np.float64(start) + np.float64(step) * np.float64(i))
        else:
            out[i] = f32(np.float64(end) - np.float64(step) * np.float64(steps - 1 - i))
    return out


def create_frustum(final_dim, downsample_factor, d_bound):
    """-> float32 [D, fH, fW, 4] = (x_pixel, y_pixel, height_bin, 1)."""
    ogfH, ogfW = final_dim
    fH, fW = ogfH // downsample_factor, ogfW // downsample_factor
    alpha = 1.5
    d_coords = np.arange(d_bound[2]) / d_bound[2]                      # float64, lss_fpn.py:333
    d_coords = np.power(d_coords, alpha)
    d_coords = d_bound[0] + d_coords * (d_bound[1] - d_bound[0])
    d_coords = d_coords.astype(f32)                                    # torch.tensor(.., float)
    D = d_coords.shape[0]
    xs = _linspace_f32(0, ogfW - 1, fW)
    ys = _linspace_f32(0, ogfH - 1, fH)
    fr = np.empty((D, fH, fW, 4), f32)
    fr[..., 0] = xs[None, None, :]
    fr[..., 1] = ys[None, :, None]
    fr[..., 2] = d_coords[:, None, None]
    fr[..., 3] = 1.0
    return fr


# --------------------------------------------------------------------------------------------
# 4x4 helpers: the build's fixed-order inverse and the sequential product
# --------------------------------------------------------------------------------------------
def inv4(A):
    """LU with partial pivoting (first max |.| wins), multipliers by reciprocal, then A X = P
    solved column by column: forward substitution (unit L, axpy form) and back substitution with
    reciprocal diagonals.  All in float32, one rounding per operation, no FMA.  Mirrors
    sgv3d_amd/csrc/geometry.hip::inv4 operation for operation."""
    A = np.array(A, dtype=f32).copy()
    n = 4
    piv = [0, 1, 2, 3]
    for j in range(n):
        p = j
        best = abs(A[j, j])
        for i in range(j + 1, n):
            if abs(A[i, j]) > best:
                best = abs(A[i, j])
                p = i
        if p != j:
            A[[j, p], :] = A[[p, j], :]
            piv[j], piv[p] = piv[p], piv[j]
        r = f32(f32(1.0) / A[j, j])
        for i in range(j + 1, n):
            A[i, j] = f32(A[i, j] * r)
        for jj in range(j + 1, n):
            for i in range(j + 1, n):
                A[i, jj] = f32(A[i, jj] - f32(A[i, j] * A[j, jj]))
    X = np.zeros((n, n), f32)
    for i in range(n):
        X[i, piv[i]] = 1.0
    dinv = [f32(f32(1.0) / A[k, k]) for k in range(n)]
    for c in range(n):
        for k in range(n):
            for i in range(k + 1, n):
                X[i, c] = f32(X[i, c] - f32(X[k, c] * A[i, k]))
        for k in range(n - 1, -1, -1):
            X[k, c] = f32(X[k, c] * dinv[k])
            for i in range(k):
                X[i, c] = f32(X[i, c] - f32(X[k, c] * A[i, k]))
    return X


def mm4(A, B):
    """C[i,j] = ((0 + a_i0 b_0j) + a_i1 b_1j) + ... in float32 (torch CPU small-bmm order)."""
    A = np.asarray(A, f32)
    B = np.asarray(B, f32)
    C = np.zeros((4, 4), f32)
    for i in range(4):
        for j in range(4):
            acc = f32(0)
            for k in range(4):
                acc = f32(acc + f32(A[i, k] * B[k, j]))
            C[i, j] = acc
    return C


def calib_prep(sensor2ego, sensor2virtual, intrin, ida, inverse=inv4):
    """Per camera 4x4 set used by the per-point pass.
    returns (ida_inv, combine_virtual, combine_ego), each float32 [4,4]."""
    ida_inv = inverse(ida)                                   # lss_fpn.py:390
    combine_virtual = mm4(sensor2virtual, inverse(intrin))   # lss_fpn.py:361
    combine_ego = mm4(sensor2ego, inverse(sensor2virtual))   # lss_fpn.py:367
    return ida_inv, combine_virtual, combine_ego


# --------------------------------------------------------------------------------------------
# G2  per-point geometry  (lss_fpn.py:350-401)
# --------------------------------------------------------------------------------------------
def _mv(M, v):
    """4x4 @ 4-vector field, sequential fp32 sum starting from 0 (torch CPU small-bmm order)."""
    out = []
    for i in range(4):
        acc = np.zeros(v[0].shape, f32)
        for k in range(4):
            with np.errstate(all='ignore'):
                acc = (acc + (M[i, k] * v[k]).astype(f32)).astype(f32)
        out.append(acc)
    return out


def geometry_points(frustum, ida_inv, combine_virtual, combine_ego, reference_height, bda=None):
    """frustum f32 [D,fH,fW,4] -> ego-frame points f32 [D,fH,fW,3] for ONE camera."""
    frustum = np.asarray(frustum, f32)
    v = [frustum[..., k] for k in range(4)]
    with np.errstate(all='ignore'):
        p = _mv(np.asarray(ida_inv, f32), v)                            # :390
        height = (f32(-1) * p[2] + f32(reference_height)).astype(f32)   # :354
        pc = [(p[0] * f32(10)).astype(f32), (p[1] * f32(10)).astype(f32),
              np.full_like(p[0], 10), p[3]]                             # :356-360
        pv = _mv(np.asarray(combine_virtual, f32), pc)                  # :362
        ratio = (height / pv[1]).astype(f32)                            # :363
        q = [(pv[k] * ratio).astype(f32) for k in range(4)]             # :365
        q[3] = np.ones_like(q[0])                                       # :366
        e = _mv(np.asarray(combine_ego, f32), q)                        # :368-369
        if bda is not None:
            e = _mv(np.asarray(bda, f32), e)                            # :394-398
    return np.stack(e[:3], -1)


# --------------------------------------------------------------------------------------------
# G3  quantise  (lss_fpn.py:487-488), GPU cast semantics
# --------------------------------------------------------------------------------------------
def cvt_i32_gpu(x):
    """float32 -> int32 like CUDA cvt.rzi.s32.f32 / AMD v_cvt_i32_f32: truncate toward zero,
    saturate to [INT32_MIN, INT32_MAX], NaN -> 0."""
    x = np.asarray(x, f32)
    out = np.zeros(x.shape, np.int32)
    nan = np.isnan(x)
    hi = x >= f32(2147483648.0)
    lo = x <= f32(-2147483648.0)
    ok = ~(nan | hi | lo)
    out[ok] = np.trunc(x[ok]).astype(np.int32)
    out[hi] = np.int32(2147483647)
    out[lo] = np.int32(-2147483648)
    return out


def voxel_params(x_bound, y_bound, z_bound):
    """voxel_size / voxel_coord / voxel_num buffers (lss_fpn.py:281-292)."""
    rows = [x_bound, y_bound, z_bound]
    voxel_size = np.array([r[2] for r in rows], f32)
    voxel_coord = np.array([r[0] + r[2] / 2.0 for r in rows], f32)
    voxel_num = np.array([int((r[1] - r[0]) / r[2]) for r in rows], np.int64)
    return voxel_size, voxel_coord, voxel_num


def quantise(points, voxel_coord, voxel_size):
    """((geom - (voxel_coord - voxel_size/2)) / voxel_size).int()  in float32."""
    voxel_coord = np.asarray(voxel_coord, f32)
    voxel_size = np.asarray(voxel_size, f32)
    origin = (voxel_coord - (voxel_size / f32(2.0)).astype(f32)).astype(f32)
    with np.errstate(all='ignore'):
        q = ((np.asarray(points, f32) - origin).astype(f32) / voxel_size).astype(f32)
    return cvt_i32_gpu(q)


def geom_xyz_for_camera(frustum, sensor2ego, sensor2virtual, intrin, ida, reference_height,
                        bda, voxel_coord, voxel_size, inverse=inv4):
    ida_inv, cv, ce = calib_prep(sensor2ego, sensor2virtual, intrin, ida, inverse=inverse)
    pts = geometry_points(frustum, ida_inv, cv, ce, reference_height, bda)
    return quantise(pts, voxel_coord, voxel_size), pts
