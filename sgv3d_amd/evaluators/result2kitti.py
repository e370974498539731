"""``kitti_evaluation`` of evaluators/result2kitti.py:62-72: read the prediction and ground-truth label folders, run the
KITTI evaluation (R40), write the result text under ``metric_path/R40`` and return the moderate 3-D AP of ``Car``."""
import os

from .kitti_utils import kitti_common as kitti
from .kitti_utils.eval import kitti_eval

__all__ = ['kitti_evaluation']


def kitti_evaluation(pred_label_path, gt_label_path, current_classes=("Car", "Pedestrian", "Cyclist"), metric_path="metric"):
    pred_annos, image_ids = kitti.get_label_annos(pred_label_path, return_ids=True)
    gt_annos = kitti.get_label_annos(gt_label_path, image_ids=image_ids)
    print(len(pred_annos), len(gt_annos))
    result, ret_dict = kitti_eval(gt_annos, pred_annos, current_classes=list(current_classes), metric="R40")
    mAP_3d_moderate = ret_dict["KITTI/Car_3D_moderate_strict"]
    os.makedirs(os.path.join(metric_path, "R40"), exist_ok=True)
    with open(os.path.join(metric_path, "R40", 'epoch_result_{}.txt'.format(round(mAP_3d_moderate, 2))), "w") as f:
        f.write(result)
    print(result)
    return mAP_3d_moderate
