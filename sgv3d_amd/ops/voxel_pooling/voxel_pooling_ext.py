"""Stand-in for the reference's pybind11 extension module ``ops.voxel_pooling.voxel_pooling_ext``.

Exports ``voxel_pooling_forward_wrapper`` with the argument order, in-place semantics and return
value of the reference's C++ wrapper (ops/voxel_pooling/src/voxel_pooling_forward.cpp:26-39,41-43),
so the reference's own ``ops/voxel_pooling/voxel_pooling.py`` runs unmodified on top of it:
copy this file and ``sgv3d_amd/_lib.py`` + ``libsgv3d_hip.so`` next to it.

Differences by design: a kernel launch failure raises ``RuntimeError`` instead of ``exit(-1)``
(voxel_pooling_forward_cuda.cu:51-55); output_features / pos_memo are checked too.
"""
import torch

from ... import _lib


def _check_input(t, name, dtype):
    # CHECK_CUDA / CHECK_CONTIGUOUS of voxel_pooling_forward.cpp:12-18; data_ptr<T>() dtype check :30-33
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDAtensor ")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous ")
    if t.dtype != dtype:
        raise RuntimeError(f"expected scalar type {dtype} but found {t.dtype}")


def voxel_pooling_forward_wrapper(batch_size, num_points, num_channels, num_voxel_x, num_voxel_y,
                                  num_voxel_z, geom_xyz_tensor, input_features_tensor,
                                  output_features_tensor, pos_memo_tensor):
    _check_input(geom_xyz_tensor, "geom_xyz_tensor", torch.int32)
    _check_input(input_features_tensor, "input_features_tensor", torch.float32)
    _check_input(output_features_tensor, "output_features_tensor", torch.float32)
    _check_input(pos_memo_tensor, "pos_memo_tensor", torch.int32)
    lib = _lib.load()
    with torch.cuda.device(input_features_tensor.device):
        rc = lib.sgv3d_voxel_pooling_forward(
            int(batch_size), int(num_points), int(num_channels),
            int(num_voxel_x), int(num_voxel_y), int(num_voxel_z),
            geom_xyz_tensor.data_ptr(), input_features_tensor.data_ptr(),
            output_features_tensor.data_ptr(), pos_memo_tensor.data_ptr(),
            _lib.stream_handle(input_features_tensor.device))
    _lib.check(rc, "voxel_pooling_forward_wrapper")
    return 1
