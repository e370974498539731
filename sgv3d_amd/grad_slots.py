"""Gradient slots: weight-gradient kernels write straight into the flat all-reduce buckets of ``train_step.FlatParams``.

Without this, a backward kernel produces ``dW`` in a fresh tensor and autograd's AccumulateGrad adds it into ``p.grad`` (a view of
the flat bucket): one small ``add`` launch per parameter, ~380 per step of the R50 model.  With it, ``FlatParams.zero_grad`` leaves
``p.grad = None`` for the parameters whose producers are known to ask for their slot, the producer (``claim``) gets a fresh view
of the bucket as its output buffer, and AccumulateGrad adopts that tensor as ``p.grad`` without a copy -- the gradient is already
where the all-reduce and the fused AdamW read it.  A second contribution to the same parameter in one step (shared weights,
gradient accumulation over micro-batches) finds the slot taken and goes through the ordinary accumulate path."""
import weakref

_SLOTS = {}          # address of the parameter's data -> [weak reference to the flat gradient bucket, offset, numel, shape, armed]
CAPABLE = set()      # addresses whose producers called claim(): zero_grad arms these


def register(param_ptr, flat_grad, offset, numel, shape):
    """The registry does not keep the bucket alive: an entry whose bucket is gone (its FlatParams was dropped) is dead."""
    _SLOTS[int(param_ptr)] = [weakref.ref(flat_grad), int(offset), int(numel), tuple(shape), False]
    CAPABLE.discard(int(param_ptr))         # (an address reused by a new parameter: its producers have to ask again)


def unregister(param_ptr):
    _SLOTS.pop(int(param_ptr), None)
    CAPABLE.discard(int(param_ptr))


def arm(param_ptr):
    s = _SLOTS.get(int(param_ptr))
    if s is not None:
        s[4] = True


def claim(param):
    """A fresh view of ``param``'s gradient slot to use as the output buffer of its gradient kernel, or None (no slot, or not
    the first contribution of this step).  The caller must fill every element and return the view as the gradient."""
    if param is None:
        return None
    key = int(param.data_ptr())
    s = _SLOTS.get(key)
    if s is None or tuple(param.shape) != s[3]:
        return None
    flat_grad = s[0]()
    if flat_grad is None:
        unregister(key)
        return None
    CAPABLE.add(key)
    if not s[4]:
        return None
    s[4] = False
    return flat_grad[s[1]:s[1] + s[2]].view(s[3])
