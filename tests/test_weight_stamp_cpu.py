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
    assert d != c
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)                      # Parameter objects replaced
    e = m._stamp()
    assert e != d
    m.load_state_dict(sd)                                   # in-place copy_: versions
    assert m._stamp() != e


def test_objects_replaced_behind_the_models_back_are_seen_on_the_next_forward():
    """no window of stale weights.  Every way torch offers to swap a tensor OBJECT under the model moves the very
    next stamp: a parent's load_state_dict(assign=True) (Lightning; it calls the children's _load_from_state_dict, never
    BEVHeight.load_state_dict), attribute assignment, a sub-module's own load_state_dict(assign=True) / .half(), a write into
    a sub-module's ``_buffers`` dict."""
    m = _model()

    class Parent(torch.nn.Module):
        def __init__(self, model):
            super().__init__()
            self.model = model
    p = Parent(m)
    a = m._stamp()
    p.load_state_dict({k: v.clone() for k, v in p.state_dict().items()}, assign=True)
    b = m._stamp()
    assert b != a
    conv = m.head.shared_conv.conv
    conv.weight = torch.nn.Parameter(conv.weight.data.clone())
    c = m._stamp()
    assert c != b
    hn = m.backbone.height_net
    hn.load_state_dict({k: v.clone() for k, v in hn.state_dict().items()}, assign=True)
    d = m._stamp()
    assert d != c
    hn.bn._buffers['running_var'] = hn.bn.running_var.clone()     # (what a sub-module's own .to() does; no hook sees it)
    e = m._stamp()
    assert e != d
    m.backbone.double()                                           # _apply of a SUB-module: BEVHeight._apply is not called
    f = m._stamp()
    assert f != e
    hn.bn.register_buffer('extra', torch.zeros(3))                # a tensor the walk did not know
    g = m._stamp()
    assert g != f and g[1] == f[1] + 1
    assert all(m._stamp() == g for _ in range(3))


def test_train_drops_the_cached_walk():
    m = _model()
    m._stamp()
    assert not m._flat_dirty
    m.train()
    assert m._flat_dirty
    m.eval()


def test_periodic_walk_catches_a_module_swapped_through_the_modules_dict():
    m = _model()
    m._RESTAMP_EVERY = 4
    a = m._stamp()
    hn = m.backbone.height_net
    import copy
    hn._modules['bn'] = copy.deepcopy(hn.bn)                # (no registration hook fires for a direct dict write)
    seen = [m._stamp() for _ in range(5)]
    assert seen[-1] != a and seen[-1][0] == a[0] + 1        # at the latest after _RESTAMP_EVERY forwards


def test_compiled_stamp_helper_equals_the_python_loop():
    """sgv3d_amd/host_ext/stamp_ext (host-only C++): the same (count, versions, addresses + ids) as the Python loop it replaces on
    the hot path of the harness's one-frame-at-a-time eval_step; through every kind of change of the test above."""
    import pytest
    from sgv3d_amd import host_ext
    fast = host_ext.stamp()
    if fast is None:
        pytest.skip("stamp_ext is not built (python -c 'import __graft_entry__ as g; g.build()')")
    m = _model()

    def both():
        a = m._stamp()
        saved, host_ext._FN = host_ext._FN, None
        try:
            b = m._stamp()
        finally:
            host_ext._FN = saved
        assert a == b
        return a
    s0 = both()
    with torch.no_grad():
        m.head.shared_conv.conv.weight.add_(1.0)
    s1 = both()
    m.backbone.height_net.bn._buffers['running_var'] = m.backbone.height_net.bn.running_var.clone()
    s2 = both()
    assert s0 != s1 != s2


def test_registration_hooks_listen_to_bevheight_trees_only():
    """The registration hooks are torch-global: they exist only while a BEVHeight instance lives, and constructing or editing an
    unrelated nn.Module does not move the counter (rounds 4-5 installed them at import and counted every registration of the
    process)."""
    import gc
    from sgv3d_amd.models import bev_height as BH
    gc.collect()
    m = _model()
    m._stamp()                                              # (the walk announces this tree's modules to the hooks)
    assert len(BH._HOOKS) == 3
    n0 = BH._REGISTRATIONS[0]
    other = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8))      # ~10 registrations, none of them ours
    other[0].weight = torch.nn.Parameter(torch.zeros(8, 3, 3, 3))
    other.register_buffer("extra", torch.zeros(1))
    assert BH._REGISTRATIONS[0] == n0
    m.head.shared_conv.conv.weight = torch.nn.Parameter(m.head.shared_conv.conv.weight.data.clone())
    assert BH._REGISTRATIONS[0] == n0 + 1                   # ours: counted
    m.backbone.add_module("probe", other)                   # an unrelated module attached to the tree: the parent announces it
    assert BH._REGISTRATIONS[0] == n0 + 2
    a = m._stamp()                                          # (the walk now tracks the attached modules too)
    other[1].register_buffer("late", torch.zeros(2))
    assert BH._REGISTRATIONS[0] == n0 + 3 and m._stamp() != a
    del m, other, a
    gc.collect()
    if not list(BH._LIVE):                                  # (other tests' models may still be alive in this process)
        assert BH._HOOKS == [] and not BH._TRACKED
