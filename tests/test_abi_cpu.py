"""CPU: the C-ABI library loads and exports exactly what include/sgv3d_hip.h declares
(no compute calls here: there is no GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "sgv3d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgv3d_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sgv3d_amd import _lib
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in sgv3d_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared, "python prototypes out of sync with the header"
    assert lib.sgv3d_abi_version() == 1


def test_conv_desc_layout_matches_header():
    from sgv3d_amd import _lib
    text = open(os.path.join(ROOT, "include", "sgv3d_hip.h")).read()
    body = text[text.index("typedef struct sgv3d_conv_desc"):text.index("} sgv3d_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in re.findall(r"int\s+([^;]+);", body):
        fields += [f.strip() for f in decl.split(",")]
    assert fields == [f[0] for f in _lib.ConvDesc._fields_]
    assert ctypes.sizeof(_lib.ConvDesc) == 4 * len(fields)


def test_pack_geometry_is_host_only():
    from sgv3d_amd import _lib
    lib = _lib.load()
    k, n = ctypes.c_int(), ctypes.c_int()
    lib.sgv3d_conv_pack_geometry(147, 64, ctypes.byref(k), ctypes.byref(n))
    assert (k.value, n.value) == (160, 128)
    lib.sgv3d_conv_pack_geometry(4608, 512, ctypes.byref(k), ctypes.byref(n))
    assert (k.value, n.value) == (4608, 512)


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any HIP call is made."""
    from sgv3d_amd import _lib
    lib = _lib.load()
    rc = lib.sgv3d_voxel_pooling_forward(0, 10, 8, 4, 4, 1, None, None, None, None, None)
    assert rc == -1 and b"non-positive" in lib.sgv3d_last_error()
    assert lib.sgv3d_voxel_plan_bytes(1, 466560, 256, 256) > 4 * (466560 + 2 * 65536)
    assert lib.sgv3d_voxel_plan_bytes(0, 1, 1, 1) == 0


def test_compiled_pybind_extension_loads_and_checks_inputs():
    """The compiled drop-in for the reference's pybind11 module (ops/voxel_pooling/src/voxel_pooling_forward.cpp:41-43):
    loads without a GPU, exports voxel_pooling_forward_wrapper with the reference's ten arguments, and refuses CPU
    tensors with the reference's message (:12-13) -- no compute call here."""
    import torch
    from sgv3d_amd.ops.voxel_pooling import compiled_ext
    ext = compiled_ext.load()
    doc = ext.voxel_pooling_forward_wrapper.__doc__
    assert doc.count("SupportsInt") == 6 and doc.count("torch.Tensor") == 4 and "-> int" in doc
    import pytest
    with pytest.raises(RuntimeError, match="must be a CUDAtensor"):
        ext.voxel_pooling_forward_wrapper(1, 2, 3, 4, 5, 1, torch.zeros(1, 2, 3, dtype=torch.int32), torch.zeros(1, 2, 3),
                                          torch.zeros(1, 5, 4, 3), torch.zeros(1, 2, 3, dtype=torch.int32))
