"""KITTI-AP evaluation of detection results on the MI355X: mirror of the reference's
``evaluators`` package for the path ``kitti_evaluation`` -> ``get_label_annos`` -> ``kitti_eval``
(evaluators/result2kitti.py:62-72).  The dataset-specific box conversion of result2kitti.py:212-393 (calibration
files of DAIR-V2X / Rope3D) is not rebuilt."""
from .result2kitti import kitti_evaluation

__all__ = ['kitti_evaluation']
