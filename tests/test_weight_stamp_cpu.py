"""Host logic of ``BEVHeight._stamp`` (no GPU): the cheap stamp between the periodic walks over the module tree sees every
way the weights can change under the model, so that packed HIP weights and captured graphs are never stale."""
import torch

from sgv3d_amd import synthetic as S
from sgv3d_amd.models.bev_height import BEVHeight


def _model():
    bc, hc = S.small_conf(depth=18)
    torch.manual_seed(0)
    return BEVHeight(bc, hc).eval()


def test_stamp_is_stable_and_sees_every_kind_of_change():
    m = _model()
    a = m._stamp()
    assert all(m._stamp() == a for _ in range(5))
    m.refresh()
    m.train(); m.eval()
    assert m._stamp() == a                                  # dropping the packed weights is not a weight change
    with torch.no_grad():
        m.head.shared_conv.conv.weight.mul_(1.0)            # in place: version counter
    b = m._stamp()
    assert b != a
    m.backbone.height_net.bn.running_mean.data = m.backbone.height_net.bn.running_mean.data.clone()   # storage swap
    c = m._stamp()
    assert c != b
    m.double()                                              # _apply: buffers become new objects
    d = m._stamp()
    assert d != c and d[0] == c[0] + 1
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)                      # Parameter objects replaced
    e = m._stamp()
    assert e != d
    m.load_state_dict(sd)                                   # in-place copy_: versions
    assert m._stamp() != e


def test_periodic_walk_catches_objects_replaced_behind_the_models_back():
    m = _model()
    m._RESTAMP_EVERY = 4
    a = m._stamp()
    bn = m.backbone.height_net.bn
    bn._buffers['running_var'] = bn.running_var.clone()     # (what a sub-module's own .to() does; no hook of ours sees it)
    seen = [m._stamp() for _ in range(5)]
    assert seen[-1] != a and seen[-1][0] == a[0] + 1        # at the latest after _RESTAMP_EVERY forwards
