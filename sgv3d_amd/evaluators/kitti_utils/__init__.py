from .eval import kitti_eval

__all__ = ['kitti_eval']
