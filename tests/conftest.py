import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the GPU suite: the hot-path rows of SURVEY §8(a)/(b) first -- the voxel_pooling operator and both of its
# boundaries, geometry / lift, the whole model, decode -- then the convolution-variant sweeps, then the §8(f) rows.  The driver
# runs `pytest -x`: a red test in a late row must not hide the parity evidence of the operator the north star names first.
_FILE_ORDER = (
    "test_voxel_pooling_gpu", "test_geometry_lift_gpu", "test_model_gpu", "test_fullsize_gpu", "test_harness_gpu",
    "test_decode_gpu", "test_closed_loop_ap_gpu",
    "test_conv_gpu", "test_conv_wino_gpu", "test_conv_bf16_gpu",
    "test_kitti_eval_gpu", "test_train_step_gpu", "test_train_forward_gpu", "test_train_head_gpu", "test_norm_grad_gpu",
    "test_conv_grad_gpu", "test_bsm_train_gpu",
)


def _file_rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    return _FILE_ORDER.index(name) if name in _FILE_ORDER else len(_FILE_ORDER)


def pytest_collection_modifyitems(config, items):
    items.sort(key=_file_rank)          # stable: the order inside a file is kept
    # a test marked gpu is skipped (not failed) when no GPU is visible, e.g. plain `pytest tests/`
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    return {name: np.load(os.path.join(GOLDEN, name + ".npz"))
            for name in ("frustum", "geometry", "voxel_pooling", "lift")}


@pytest.fixture(scope="session")
def hip():
    """The product library; the GPU tests call the kernels through this C ABI."""
    from sgv3d_amd import _lib
    _lib.load()
    return _lib
