"""KITTI-AP evaluation of detection results on the MI355X: mirror of the reference's
``evaluators`` package for the path ``kitti_evaluation`` -> ``get_label_annos`` -> ``kitti_eval``
(evaluators/result2kitti.py:62-72), the detection -> label-file conversion ``result2kitti`` for the KITTI-format data roots
(:212-268) and ``RoadSideEvaluator`` (evaluators/det_evaluators.py) tying them to the model's ``get_bboxes`` output."""
from .det_evaluators import RoadSideEvaluator
from .result2kitti import kitti_evaluation, result2kitti, result2kitti_dair

__all__ = ['RoadSideEvaluator', 'kitti_evaluation', 'result2kitti', 'result2kitti_dair']
