"""``BEVHeight`` — the detector facade, drop-in for the reference's models/bev_height.py:11-126.

Same constructor (``backbone_conf``, ``head_conf``, ``is_train_height``, ``checkpoint``), same
``forward(x, mats_dict, timestamps=None)`` return structure, same attribute paths
(``.backbone.img_backbone`` ..., ``.head``) and parameter names, so the reference's Lightning
module can do ``self.model = BEVHeight(self.backbone_conf, self.head_conf)`` unchanged
(exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:209) and load its checkpoints.

The forward is one stream of hand-written gfx950 kernels; between backbone and head the BEV map
stays in the NHWC buffer voxel pooling produced (the reference permutes it to NCHW and makes it
contiguous, layers/backbones/lss_fpn.py:495, only for cuDNN's benefit).
"""
import torch
from torch import nn

from ..layers.backbones.bsm_lss_fpn import BSMLSSFPN
from ..layers.backbones.lss_fpn import LSSFPN
from ..layers.blocks import HipModule
from ..layers.heads.bev_height_head import BEVHeightHead

__all__ = ['BEVHeight']


class BEVHeight(nn.Module):
    """Detector = camera backbone (``LSSFPN`` or, with ``backbone_conf['is_bsm']``, ``BSMLSSFPN``) + BEV head.

    ``backbone_conf`` / ``head_conf`` are the experiment files' dicts, taken verbatim; ``is_train_height``
    makes the training-mode forward return ``(preds, height_pred)`` as models/bev_height.py:72-77 does;
    ``checkpoint`` optionally names a Lightning checkpoint whose ``model.backbone.*`` entries initialise the
    backbone."""

    def __init__(self, backbone_conf, head_conf, is_train_height=False, checkpoint=None):
        super(BEVHeight, self).__init__()
        if backbone_conf['is_bsm']:
            self.backbone = BSMLSSFPN(**backbone_conf)
        else:
            self.backbone = LSSFPN(**backbone_conf)
        self.head = BEVHeightHead(**head_conf)
        self.is_train_height = is_train_height
        self._param_stamp = None
        if checkpoint is not None:
            with open(checkpoint, "rb") as f:
                state_dict = torch.load(f, map_location='cpu')
            self.backbone.load_state_dict(self.get_backbone(state_dict))

    @staticmethod
    def get_backbone(state_dict):
        """Backbone sub-dict of a Lightning checkpoint ('model.backbone.' prefix stripped).  The
        reference's version (models/bev_height.py:35-40) lacks ``self`` and keeps a leading '.'."""
        state_dict = state_dict.get('state_dict', state_dict)
        prefix = 'model.backbone.'
        return {k[len(prefix):]: v for k, v in state_dict.items() if k.startswith(prefix)}

    # ------------------------------------------------------------------------------------------
    def _stamp(self):
        return tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def refresh(self):
        """Drop the packed HIP weights (they are rebuilt on the next forward).  Called automatically
        when a parameter or buffer changed in place or was replaced (load_state_dict, .to(), optimiser)."""
        for m in self.modules():
            if isinstance(m, HipModule):
                m._hip = None
        self._param_stamp = None

    def train(self, mode=True):
        """Switching between training and inference drops the packed inference weights: the fused optimiser step
        (train_step.DataParallelAdamW) writes parameters through raw pointers, which no version counter records."""
        super().train(mode)
        self.refresh()
        return self

    def forward(self, x, mats_dict, timestamps=None):
        """Images -> per-task prediction maps, as models/bev_height.py:42-80 does.

        ``x``: float32 [B, num_sweeps, num_cams, 3, H, W] on the GPU.  ``mats_dict``: the seven calibration
        tensors of the reference's collate function -- 'sensor2ego_mats', 'intrin_mats', 'ida_mats',
        'sensor2sensor_mats', 'sensor2virtual_mats' as [B, num_sweeps, num_cams, 4, 4], 'reference_heights' as
        [B, num_sweeps, num_cams], 'bda_mat' as [B, 4, 4].  ``timestamps`` is ignored here as it is there.
        Result: one single-element list per task holding a dict of NCHW maps (reg, height, dim, rot, vel,
        heatmap) -- the nesting mmdet3d's CenterHead produces."""
        if self.training:
            from ..train_forward import bevheight_train_forward
            return bevheight_train_forward(self, x, mats_dict)          # differentiable
        stamp = self._stamp()
        if stamp != self._param_stamp:
            self.refresh()
            self._param_stamp = stamp
        bev = self.backbone(x, mats_dict, timestamps, nhwc_out=True)   # NHWC buffer [B, Y, X, C]
        return self.head(bev, nhwc=True)

    def get_targets(self, gt_boxes, gt_labels):
        return self.head.get_targets(gt_boxes, gt_labels)

    def loss(self, targets, preds_dicts):
        return self.head.loss(targets, preds_dicts)

    def get_bboxes(self, preds_dicts, img_metas=None, img=None, rescale=False):
        return self.head.get_bboxes(preds_dicts, img_metas, img, rescale)
