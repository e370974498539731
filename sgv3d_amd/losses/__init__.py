"""Mirror of the reference's ``losses`` package for the pieces the SGV3D configs use (losses/__init__.py:1-4):
``FocalLoss`` on the MI355X (csrc/bsm_train.hip) plus the semantic supervision of the BSM experiment."""
from .focal import BINARY_MODE, MULTICLASS_MODE, MULTILABEL_MODE, FocalLoss, focal_loss_with_logits
from .semantic import SemanticSupervision, downsample_gt_semantic

__all__ = ['BINARY_MODE', 'MULTICLASS_MODE', 'MULTILABEL_MODE', 'FocalLoss', 'focal_loss_with_logits',
           'SemanticSupervision', 'downsample_gt_semantic']
