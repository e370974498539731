"""Training-step semantics the reference's configs ask for, checked without a GPU:
``frozen_stages=0`` of the image backbone (exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:48; mmdet 2.19.0
``ResNet._freeze_stages`` / ``ResNet.train``) in the product's module tree and in the oracle's training-mode forward."""
import torch

from oracle import torch_model as O
from sgv3d_amd import synthetic
from sgv3d_amd.layers.blocks import ResNet
from sgv3d_amd.models.bev_height import BEVHeight


def _flags(r):
    return (r.bn1.training, r.conv1.weight.requires_grad, r.bn1.weight.requires_grad, r.bn1.bias.requires_grad)


def test_resnet_freeze_stages_follows_mmdet():
    r = ResNet(depth=18, frozen_stages=0, norm_eval=False)
    assert _flags(r) == (False, False, False, False)                      # frozen by the constructor, as upstream
    assert r.layer1[0].bn1.training and all(p.requires_grad for p in r.layer1.parameters())
    r.eval()
    assert not r.layer1[0].bn1.training
    r.train()
    assert _flags(r) == (False, False, False, False) and r.layer1[0].bn1.training     # train() re-freezes the stem only
    assert r.frozen_stem()
    r2 = ResNet(depth=18, frozen_stages=2, norm_eval=False).train()
    assert not any(p.requires_grad for p in r2.layer1.parameters()) and not any(p.requires_grad for p in r2.layer2.parameters())
    assert not any(m.training for m in r2.layer2.modules()) and all(p.requires_grad for p in r2.layer3.parameters())
    assert r2.layer3[0].bn1.training
    r3 = ResNet(depth=18, frozen_stages=-1, norm_eval=False).train()
    assert _flags(r3) == (True, True, True, True) and not r3.frozen_stem()
    r4 = ResNet(depth=18, frozen_stages=-1, norm_eval=True).train()
    assert not any(m.training for m in r4.modules() if isinstance(m, torch.nn.BatchNorm2d))
    assert all(p.requires_grad for p in r4.parameters())                  # norm_eval keeps the affine parameters trainable


def test_detector_train_mode_freezes_the_image_stem_only():
    bconf, hconf = synthetic.small_conf()
    model = BEVHeight(bconf, hconf).train()
    r = model.backbone.img_backbone
    assert _flags(r) == (False, False, False, False)
    assert model.head.trunk.bn1.training and model.head.trunk.conv1.weight.requires_grad      # the BEV trunk has no frozen stage (:77-87)
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    assert sorted(frozen) == ['backbone.img_backbone.bn1.bias', 'backbone.img_backbone.bn1.weight', 'backbone.img_backbone.conv1.weight']


def test_oracle_training_forward_honours_frozen_stages():
    """The oracle's training-mode forward with ``frozen_stages=0`` reads bn1's running statistics (its output moves with them) and
    leaves conv1 / bn1 without gradient when the caller, like the tests, marks only trainable parameters; with ``frozen_stages=-1``
    bn1 runs on batch statistics (the running ones do not matter)."""
    bconf, hconf = synthetic.small_conf(depth=18)
    torch.manual_seed(0)
    model = BEVHeight(bconf, hconf)
    synthetic.randomize_norm_stats_(model, seed=3)
    imgs = synthetic.make_images(2, final=bconf['final_dim'], seed=1)
    mats = synthetic.make_mats(2, scale=bconf['final_dim'][0] / 864)
    assert O.frozen_prefixes('backbone.img_backbone', bconf['img_backbone_conf']) == ('backbone.img_backbone.conv1.', 'backbone.img_backbone.bn1.')

    def heat(frozen, shift):
        bc = dict(bconf, img_backbone_conf=dict(bconf['img_backbone_conf'], frozen_stages=frozen))
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        sd['backbone.img_backbone.bn1.running_mean'] += shift
        names = [n for n, p in model.named_parameters() if p.requires_grad or frozen < 0]
        for n in names:
            sd[n].requires_grad_(True)
        preds = O.bevheight_train_forward(sd, bc, hconf, imgs, mats)
        sum(p[0]['heatmap'].square().sum() for p in preds).backward()
        return preds[0][0]['heatmap'].detach(), sd

    a, sd_a = heat(0, 0.0)
    b, _ = heat(0, 0.5)
    assert float((a - b).abs().max()) > 0                               # eval-mode bn1: the running mean is an input
    assert sd_a['backbone.img_backbone.conv1.weight'].grad is None and sd_a['backbone.img_backbone.bn1.weight'].grad is None
    assert sd_a['backbone.img_backbone.layer1.0.conv1.weight'].grad is not None
    c, sd_c = heat(-1, 0.0)
    d, _ = heat(-1, 0.5)
    assert torch.equal(c, d)                                            # batch statistics: the running mean is not read
    assert sd_c['backbone.img_backbone.conv1.weight'].grad is not None
    assert float((a - c).abs().max()) > 0
