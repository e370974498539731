"""sgv3d_amd — MI355X (gfx950) native camera->BEV forward path for SGV3D / BEVHeight.

Package layout mirrors the reference's interface for this path and nothing else:

* ``sgv3d_amd.ops.voxel_pooling``          <- reference ``ops/voxel_pooling`` (operator boundary)
* ``sgv3d_amd.layers.backbones.lss_fpn``   <- reference ``layers/backbones/lss_fpn.py``
* ``sgv3d_amd.layers.heads.bev_height_head`` <- reference ``layers/heads/bev_height_head.py``
* ``sgv3d_amd.models.bev_height``          <- reference ``models/bev_height.py`` (module boundary)
* ``sgv3d_amd.csrc``                       HIP kernels + the C ABI (``include/sgv3d_hip.h``)

Everything computes on the GPU through ``libsgv3d_hip.so``; there is no CPU fallback.
"""
__version__ = "0.1.0"
