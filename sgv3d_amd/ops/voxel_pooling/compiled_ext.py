"""Loader of the COMPILED pybind11 module ``voxel_pooling_ext`` (src/voxel_pooling_ext.cpp, built by src/Makefile /
``__graft_entry__.build()``): the drop-in for the reference's ``ops/voxel_pooling/voxel_pooling_ext.cpython-*.so``
(ops/voxel_pooling/src/voxel_pooling_forward.cpp:41-43).  ``voxel_pooling_ext.py`` next to this file is the ctypes
equivalent; both bind ``sgv3d_voxel_pooling_forward`` of libsgv3d_hip.so.  A maintainer of the reference copies the
built ``.so`` (and libsgv3d_hip.so) into ``ops/voxel_pooling/`` and changes nothing else."""
import glob
import importlib.util
import os

_MOD = None


def path():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "src")
    found = sorted(glob.glob(os.path.join(here, "voxel_pooling_ext*.so")))
    if not found:
        raise RuntimeError(f"compiled voxel_pooling_ext not found under {here}: run `make -C {here}` "
                           "(or `python -c 'import __graft_entry__ as g; g.build()'`)")
    return found[-1]


def load():
    global _MOD
    if _MOD is None:
        import torch  # noqa: F401   (the module links libtorch / libc10_hip)
        spec = importlib.util.spec_from_file_location("voxel_pooling_ext", path())
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _MOD = mod
    return _MOD
