"""Drop-in for the reference package ``ops/voxel_pooling`` (ops/voxel_pooling/__init__.py:1-3)."""
from .voxel_pooling import voxel_pooling, VoxelPooling, VoxelPlan, set_mode, get_mode

__all__ = ['voxel_pooling', 'VoxelPooling', 'VoxelPlan', 'set_mode', 'get_mode']
