"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's algorithm for the camera->BEV forward path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the
product (``sgv3d_amd``) never does.

Pinning status (see DESIGN.md "Oracle"):
* voxel pooling, geometry, frustum, lift: pinned against golden vectors produced by executing the
  reference's own Python in the build container (tests/golden/make_golden.py).
* conv / BN / ResNet / SECONDFPN / DCN / CenterHead restatements (oracle/torch_model.py): the
  arithmetic lives in third-party wheels (mmcv-full 1.4.0, mmdet 2.19.0, mmdet3d 0.18.1) that are
  absent from /root/reference and from this image, and the reference has no tests for them:
  PARITY UNPINNED for those layers (restated from the published definitions; structural
  known-answer tests only).
"""
