"""Data-parallel optimiser step (BASELINE config 4: global batch 32 over 8 MI355X).

The reference wraps the model in Lightning's DDP (exps/base_cli.py:57-58 ``accelerator='ddp'``) and steps
``torch.optim.AdamW(lr = 2e-4 / 64 * batch * gpus, weight_decay=1e-7)`` under ``MultiStepLR([19, 23])``
(exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:197,298-305).  Here:

* ``FlatParams`` re-homes every parameter and its gradient as views of a few large flat fp32 buffers (buckets) in
  reverse registration order -- the order backward produces gradients in -- so a bucket is ONE contiguous RCCL
  all-reduce and ONE fused AdamW launch.  Buckets default to 48 MiB (``DEFAULT_BUCKET_BYTES``): xGMI is point-to-point
  (7 links, ~153 GB/s each), a ring all-reduce is bound by one link -- 48 MiB is ~0.6 ms of wire time, far above the
  collective's launch latency --, and the ~0.3 GB of BEVHeight-R50 gradients become 7 collectives of which the first leaves
  when the CenterHead's and the BEV trunk's gradients are done, with ~85 % of backward still to run under it (256 MiB
  buckets, the earlier default, held the first collective back until backward was almost over).
* ``all_reduce_grads`` launches the sum all-reduces asynchronously on torch's RCCL stream (``async_op=True``) as soon
  as it is called per bucket; the 1 / world factor is folded into the AdamW kernel (no averaging pass over HBM).
* ``step`` waits for each bucket's collective and runs ``sgv3d_adamw_step`` on it.
* ``max_grad_norm`` (Lightning's ``gradient_clip_val=5`` of the reference's Trainer, exps/...:405): the global L2 norm of the
  averaged gradient over all buckets -- one ``sgv3d_grad_sumsq`` per bucket, one ``sgv3d_clip_coef`` -- and the coefficient
  ``min(1, max_norm / (norm + 1e-6))`` of ``torch.nn.utils.clip_grad_norm_`` read by the AdamW kernel from device memory:
  no pass that rewrites the gradients, no host round trip, capturable.

* at construction with more than one rank the flat parameter buckets are broadcast from rank 0 (what Lightning's DDP
  does when it wraps the model), so replicas start identical whatever each rank's seed was; ``check_replicas`` compares
  a checksum of the parameters over the ranks and raises on drift.

* the weight-gradient and BatchNorm-gradient kernels write into the buckets directly (``grad_slots``): ``zero_grad`` leaves
  ``p.grad = None`` for those parameters, the kernel's output is a view of the bucket and autograd adopts it as ``p.grad``
  (no accumulate ``add`` per parameter; ~380 launches per step of the R50 model).

Deviation from ``torch.optim.AdamW``: gradients live as always-defined, zero-filled views of the buckets, so a parameter
that received no gradient in a step (``assist_layer`` without ``is_train_height``, lss_fpn.py:459,493-495) still gets
its moments decayed and its weight decay applied, where torch skips parameters whose ``.grad`` is None.  With the
reference's weight_decay of 1e-7 the difference is below fp32 resolution per step.

With the "gloo" backend and CPU tensors (the world-size-2 tests) the collectives run through gloo; the fused update
itself needs the GPU library and raises without it.
"""
import os

import torch

from . import _lib, grad_slots, pack_cache

DIRECT_GRADS = os.environ.get("SGV3D_DIRECT_GRADS", "1") != "0"   # 0: every gradient goes through autograd's accumulate add

DEFAULT_BUCKET_BYTES = int(os.environ.get("SGV3D_BUCKET_MIB", "48")) << 20
REFERENCE_GRADIENT_CLIP_VAL = 5.0     # exps/bevheight/dair-v2x/bev_height_lss_r50_864_1536_256x256.py:405, exps/sgv3d/...r101...:529

__all__ = ['FlatParams', 'DataParallelAdamW', 'GraphedTrainStep', 'reference_lr', 'multistep_lr', 'DEFAULT_BUCKET_BYTES',
           'REFERENCE_GRADIENT_CLIP_VAL']


def reference_lr(batch_size_per_device, gpus, basic_lr_per_img=2e-4 / 64):
    """exps/...:197,299: lr = basic_lr_per_img * batch_size_per_device * gpus."""
    return basic_lr_per_img * batch_size_per_device * gpus


def multistep_lr(base_lr, epoch, milestones=(19, 23), gamma=0.1):
    """torch.optim.lr_scheduler.MultiStepLR as a function of the epoch (exps/...:303)."""
    return base_lr * gamma ** sum(1 for m in milestones if epoch >= m)


class FlatParams:
    def __init__(self, params, bucket_bytes=None):
        bucket_bytes = DEFAULT_BUCKET_BYTES if bucket_bytes is None else int(bucket_bytes)
        params = [p for p in params if p.requires_grad]     # (frozen_stages: the image backbone's stem is not in any bucket)
        assert params, "no trainable parameters"
        assert all(p.dtype == torch.float32 for p in params), "fp32 parameters only"
        self.params = params
        dev = params[0].device
        order = list(reversed(params))                 # backward reaches the last layers first
        self.buckets = []                              # (flat_param, flat_grad, [(param, offset, numel)])
        cur, cur_n = [], 0
        limit = max(1, bucket_bytes // 4)
        for p in order:
            n = (p.numel() + 3) // 4 * 4               # keep every view 16-byte aligned
            if cur and cur_n + n > limit:
                self._close(cur, cur_n, dev)
                cur, cur_n = [], 0
            cur.append((p, cur_n, p.numel()))
            cur_n += n
        self._close(cur, cur_n, dev)

    def _close(self, entries, n, dev):
        flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off, cnt in entries:
            flat_p[off:off + cnt].copy_(p.data.reshape(-1))
            p.data = flat_p[off:off + cnt].view(p.shape)
            p.grad = flat_g[off:off + cnt].view(p.shape)
            grad_slots.register(p.data_ptr(), flat_g, off, cnt, p.shape)
        self.buckets.append((flat_p, flat_g, entries))

    def zero_grad(self):
        """Zero the buckets.  Parameters whose gradient kernels write into the bucket themselves (grad_slots) get
        ``p.grad = None``: the kernel's output view is adopted by autograd as ``p.grad``, no accumulate launch."""
        for _, g, entries in self.buckets:
            g.zero_()
            if grad_slots.CAPABLE and DIRECT_GRADS:
                for p, off, cnt in entries:
                    if p.data_ptr() in grad_slots.CAPABLE:
                        p.grad = None
                        grad_slots.arm(p.data_ptr())

    def check_views(self):
        """Gradients must still live in the flat buffers (an optimiser / autograd that replaced ``p.grad`` would
        silently cut the bucket out of the all-reduce)."""
        for flat_p, flat_g, entries in self.buckets:
            for p, off, cnt in entries:
                if p.grad is None and p.data_ptr() in grad_slots.CAPABLE:
                    continue                    # no contribution this step: the slot holds the zeros of zero_grad
                if p.grad is None or p.grad.data_ptr() != flat_g.data_ptr() + off * 4:
                    raise _lib.SGV3DError("a parameter's .grad no longer aliases its bucket; use zero_grad() of FlatParams")
                if p.data_ptr() != flat_p.data_ptr() + off * 4:
                    raise _lib.SGV3DError("a parameter no longer aliases its bucket")


class DataParallelAdamW:
    """AdamW over ``FlatParams`` with the gradient all-reduce of the data-parallel step."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-7, bucket_bytes=None, group=None,
                 max_grad_norm=None):
        self.flat = params if isinstance(params, FlatParams) else FlatParams(params, bucket_bytes)
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        self.group = group
        self.max_grad_norm = None if not max_grad_norm else float(max_grad_norm)     # None / 0: no clipping (Lightning's default)
        self._clip = None             # device [coefficient, total norm] of the step that ran last
        self._partials = None
        self.state = [(torch.zeros_like(p), torch.zeros_like(p)) for p, _, _ in self.flat.buckets]
        self.steps = 0
        self._pending = []
        self._hyper = None            # device [lr, lr / bc1, 1 / sqrt(bc2)] of the step about to run (GraphedTrainStep)
        self._in_graph = False        # inside GraphedTrainStep's capture: no collectives from the gradient hooks
        self.first_early_event = None # a torch.cuda.Event recorded on the backward stream when bucket 0's all-reduce is launched
        self.packs = None             # pack_cache.PackCache of these parameters (made by the first zero_grad)
        if self._collectives():
            self.broadcast_parameters()

    def broadcast_parameters(self, src=0):
        """Every rank takes rank ``src``'s parameters (one broadcast per flat bucket)."""
        import torch.distributed as dist
        for p, _, _ in self.flat.buckets:
            dist.broadcast(p, src=src, group=self.group)

    def check_replicas(self):
        """Raise if the parameters differ between ranks (sum and sum of squares of every bucket, compared through one
        MIN and one MAX all-reduce).  Cheap enough to call every few hundred steps."""
        import torch.distributed as dist
        if not self._collectives():
            return
        sums = torch.stack([torch.stack([p.double().sum(), (p.double() ** 2).sum()]) for p, _, _ in self.flat.buckets])
        lo, hi = sums.clone(), sums.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if not torch.equal(lo, hi):
            raise _lib.SGV3DError("data-parallel replicas have drifted apart: parameter checksums differ between ranks")

    def _world(self):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group)
        return 1

    def _collectives(self):
        """True when the step talks to the process group: more than one rank, or a 1-rank group with
        SGV3D_FORCE_DIST=1 (runs broadcast / all-reduce through RCCL on a single-GPU box: the stand-in for BASELINE
        configs[3], whose 8-GPU node only the driver has)."""
        import os
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        return dist.get_world_size(self.group) > 1 or bool(os.environ.get("SGV3D_FORCE_DIST"))

    def zero_grad(self):
        self.flat.zero_grad()
        if hasattr(self, '_left') and not self._pending:
            self._arm()
        # the step's window of kept packed weights opens here: every registered packed form is rewritten from the parameters as they
        # are now (one launch), and stays valid until step() changes them (sgv3d_amd/pack_cache.py)
        if pack_cache.ENABLED and self.flat.buckets[0][0].is_cuda:
            if self.packs is None:
                self.packs = pack_cache.PackCache(self.flat.params)
            pack_cache.ACTIVE = self.packs
            self.packs.refresh(self.flat.buckets[0][0].device)

    def overlap_with_backward(self):
        """Launch a bucket's all-reduce from inside backward, as soon as the last of its parameters has accumulated its
        gradient (post-accumulate-grad hooks; buckets are filled in backward order, so the first collectives run under
        the rest of the backward pass like DDP's).  A bucket holding a parameter that receives no gradient in a step
        never completes in the first step (``all_reduce_grads`` / ``step`` launch whatever is still missing); the set of such
        parameters is learned from that step and not waited for afterwards (every rank sees the same set: it follows from
        the model's flags, not from the data).

        Contract: ONE backward per optimiser step.  The per-bucket counters are re-armed by ``zero_grad`` /
        ``all_reduce_grads`` / ``step`` (each of them, so any of the usual loop shapes works); a second backward before the step (gradient accumulation) would add local
        gradients on top of an already reduced bucket, so it raises instead.  Early collectives are launched in
        bucket order only (bucket i after bucket i-1): every rank issues the same sequence of collectives even if the
        set of parameters that receive a gradient differs between ranks (a bucket that completes out of order waits
        for ``all_reduce_grads``)."""
        self._hooks = []
        self._unused = set()          # ids of parameters that got no gradient in the first step (learned once)
        self._learned = False
        self._fired = set()
        self._arm()
        for bi, (_, _, entries) in enumerate(self.flat.buckets):
            for p, _, _ in entries:
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, bi=bi: self._on_grad(bi, id(_p))))
        return self

    def _arm(self):
        self._early = {}
        self._left = [sum(1 for p, _, _ in entries if id(p) not in self._unused) for _, _, entries in self.flat.buckets]
        self._hold = set()            # buckets kept for all_reduce_grads this step (an "unused" parameter fired after all)
        self._next_early = 0
        self._fired = set()

    def _on_grad(self, bi, pid):
        import torch.distributed as dist
        self._fired.add(pid)
        if pid in self._unused:
            # not counted in _left: a parameter that had no gradient in the first step (assist_layer without
            # is_train_height, lss_fpn.py:459,493-495) got one now
            if bi in self._early:
                raise _lib.SGV3DError("overlap_with_backward: a parameter that received no gradient in the first step got one "
                                      "after its bucket's all-reduce was launched; call overlap_with_backward() again after "
                                      "changing which parameters train")
            self._hold.add(bi)
            return
        if self._in_graph:
            return
        self._left[bi] -= 1
        if self._left[bi] < 0:
            raise _lib.SGV3DError("overlap_with_backward: a second backward ran before step(); the bucket all-reduces of "
                                  "the first one are already in flight (accumulate micro-batches without the overlap, or "
                                  "call step() / zero_grad() between backwards)")
        if self._collectives():
            # fixed order: launch every complete bucket from the front of the queue
            while (self._next_early < len(self._left) and self._left[self._next_early] == 0
                   and self._next_early not in self._hold):
                i = self._next_early
                if i == 0 and self.first_early_event is not None:
                    self.first_early_event.record()           # (diagnostic: where in backward the first collective leaves)
                self._early[i] = dist.all_reduce(self.flat.buckets[i][1], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self._next_early += 1

    def all_reduce_grads(self):
        """Start the sum all-reduce of every bucket that is not already in flight (asynchronous; ``step`` waits per
        bucket)."""
        import torch.distributed as dist
        self._pending = []
        early = getattr(self, '_early', {})
        if self._collectives():
            for bi, (_, g, _) in enumerate(self.flat.buckets):
                self._pending.append(early[bi] if bi in early else
                                     dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if hasattr(self, '_left'):
            self._learn_unused()
            self._arm()

    def _learn_unused(self):
        if not self._learned and self._fired:
            # parameters without a gradient in the first step never complete their bucket (DDP's "unused parameters"):
            # from now on they are not waited for.  Their (zero) gradient views still travel with the bucket.
            self._unused = {id(p) for _, _, entries in self.flat.buckets for p, _, _ in entries} - self._fired
            self._learned = True

    def stage_hyper(self, lr=None):
        """Advance the step counter and put the scalars of that step -- [lr, lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)], computed as
        sgv3d_adamw_step computes them -- into the device buffer the recorded update reads: one ``sgv3d_adamw_set_hyper`` launch on the
        current stream whose KERNEL ARGUMENTS carry the three numbers (copied at launch), so staging step t + 1 while the replay of
        step t is still queued cannot disturb step t (a pinned staging buffer rewritten by the host could).
        ``step(recorded=True)`` launches the update against that buffer."""
        lr = self.lr if lr is None else float(lr)
        self.steps += 1
        dev = self.flat.buckets[0][0].device
        if self._hyper is None:
            self._hyper = torch.zeros(4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.load().sgv3d_adamw_set_hyper(self._hyper.data_ptr(), self.steps, lr, self.betas[0], self.betas[1],
                                                   _lib.stream_handle(dev))
        _lib.check(rc, "sgv3d_adamw_set_hyper")

    def grad_norm(self):
        """Total L2 norm of the averaged gradient of the last step (before clipping; one device -> host read).  None without
        ``max_grad_norm``."""
        return None if self._clip is None else float(self._clip[1])

    def clip_coefficient(self):
        return None if self._clip is None else float(self._clip[0])

    def _clip_coef(self, world):
        """Enqueue the global-norm coefficient of the gradients now in the buckets -> device pointer for the AdamW launches."""
        lib = _lib.load()
        dev = self.flat.buckets[0][0].device
        if not self.flat.buckets[0][0].is_cuda:
            raise _lib.SGV3DError("gradient clipping runs on the GPU (no CPU fallback)")
        per = lib.sgv3d_grad_sumsq_partials()
        if self._clip is None:
            self._clip = torch.zeros(2, dtype=torch.float32, device=dev)
            self._partials = torch.zeros(per * len(self.flat.buckets), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            st = _lib.stream_handle(dev)
            for i, (_, g, _) in enumerate(self.flat.buckets):
                _lib.check(lib.sgv3d_grad_sumsq(g.numel(), g.data_ptr(), self._partials.data_ptr() + i * per * 8, st), "sgv3d_grad_sumsq")
            _lib.check(lib.sgv3d_clip_coef(self._partials.data_ptr(), per * len(self.flat.buckets), 1.0 / world, self.max_grad_norm,
                                           self._clip.data_ptr(), st), "sgv3d_clip_coef")
        return self._clip.data_ptr()

    def step(self, lr=None, recorded=False):
        """One AdamW update of every bucket with the averaged gradients (call ``all_reduce_grads`` first when the
        process group has more than one rank).  ``recorded``: the step-dependent scalars come from the device buffer that
        ``stage_hyper`` filled (the launch can sit in a hipGraph); the step counter is then ``stage_hyper``'s to advance."""
        lr = self.lr if lr is None else float(lr)
        world = self._world()
        if self.packs is not None:
            self.packs.close()          # the weights change below: packed forms are stale until the next zero_grad refreshes them
        if self._collectives() and not self._pending and not self._in_graph:
            self.all_reduce_grads()
        self.flat.check_views()
        if not recorded:
            self.steps += 1
        elif self._hyper is None:
            raise _lib.SGV3DError("step(recorded=True) needs stage_hyper() first")
        lib = _lib.load()
        clip = None
        if self.max_grad_norm is not None:
            # the norm is over ALL buckets: every collective has to be in before the first update
            for w in self._pending:
                w.wait()
            self._pending = []
            clip = self._clip_coef(world)
        for i, ((p, g, _), (m, v)) in enumerate(zip(self.flat.buckets, self.state)):
            if self._pending:
                self._pending[i].wait()
            if not p.is_cuda:
                raise _lib.SGV3DError("the fused AdamW update runs on the GPU (no CPU fallback)")
            with torch.cuda.device(p.device):
                if recorded:
                    rc = lib.sgv3d_adamw_step_dev(p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), self._hyper.data_ptr(),
                                                  self.betas[0], self.betas[1], self.eps, self.weight_decay, 1.0 / world, clip,
                                                  _lib.stream_handle(p.device))
                else:
                    rc = lib.sgv3d_adamw_step(p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), self.steps, lr,
                                              self.betas[0], self.betas[1], self.eps, self.weight_decay, 1.0 / world, clip,
                                              _lib.stream_handle(p.device))
            _lib.check(rc, "sgv3d_adamw_step")
        self._pending = []
        if hasattr(self, '_left') and self._fired:
            # overlap hooks installed and a backward ran since the counters were last armed (the no-collectives path never
            # goes through all_reduce_grads, and a loop may zero gradients with model.zero_grad() instead of ours): learn the
            # unused-parameter set and re-arm here, so that the next backward is this step's successor, not a "second backward"
            self._learn_unused()
            self._arm()


class GraphedTrainStep:
    """One optimiser step -- ``opt.zero_grad()``, ``loss = forward_backward()``, the fused AdamW update -- recorded once as a
    hipGraph and replayed per step.  The eager step of the R50 model is ~1500 launches from Python and the autograd engine; on a
    host that needs 35 us per launch the GPU (45 ms of kernels at batch 2) waits for the launches, not the other way round.

    ``forward_backward``: a callable without arguments that runs forward, loss and ``loss.backward()`` on STATIC tensors (the
    caller refreshes them in place -- ``imgs.copy_(next_imgs)`` ... -- before each call; ground-truth boxes padded to a fixed
    count with label -1, which ``BEVHeightHead.get_targets`` ignores) and returns the loss tensor.  Everything it launches must be
    capturable: no ``.item()`` / ``.tolist()`` / host-side shape decisions on device data.  The per-layer kernel choices must
    exist already (run a few eager steps first, as any warm-up does).

    What the replay does not do: Python.  Hooks, ``p.grad`` re-binding, logging inside ``forward_backward`` run at capture time
    only; the gradients live where the capture left them (views of the flat buckets), BatchNorm statistics, step counters and
    the dropout stream advance on the device as in eager steps.

    With a process group (data parallel): the recorded part ends after backward; the bucket all-reduces and the AdamW update are
    launched eagerly after each replay (5 launches), so that no collective is part of a graph.  The loss's own all-reduce of its
    averaging factors (``BEVHeightHead.loss``) has to be capturable by the backend (RCCL is; gloo is not) -- ``capture_error``
    holds the exception if it was not, and the object then runs the eager step.

    Construction runs ``warmup`` REAL eager steps on the capture stream first (allocator pools, lazily created state), then records;
    the recorded step itself first runs at the first call.  ``graphed()`` returns the (static) loss tensor; ``graphed.result`` is the
    same tensor."""

    def __init__(self, forward_backward, opt, lr=None, warmup=1, strict=False):
        self.fn, self.opt = forward_backward, opt
        self.result = None
        self.graph = None
        self.capture_error = None
        self.replays = 0
        self.in_graph_update = not opt._collectives()
        dev = opt.flat.buckets[0][0].device
        from .pipeline import capture_begin, CAPTURE_ERRORS
        # autograd keeps a parameter's AccumulateGrad node -- and the stream it was created on -- alive while a post-accumulate hook is
        # registered on it (overlap_with_backward); a backward on the capture stream would then synchronise with that older stream,
        # which a capture cannot record.  The hooks are not needed while the graph runs (no Python in a replay; the all-reduces are
        # launched after it): removed here, re-installed if the capture fails.  (References of the CALLER to an earlier loss tensor keep
        # the nodes alive the same way: drop them -- ``loss = float(loss)`` -- before constructing this object.)
        had_hooks = bool(getattr(opt, '_hooks', None))
        if had_hooks:
            for h in opt._hooks:
                h.remove()
            opt._hooks = []
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(0, int(warmup))):      # on the capture stream: allocator pools, lazily created state
                self._eager(lr)
            side.synchronize()
            g = torch.cuda.CUDAGraph()
            steps_before = opt.steps
            try:
                opt._in_graph = True
                if self.in_graph_update:
                    opt.stage_hyper(lr)               # (outside the graph: one launch per step whose arguments carry the scalars)
                # with a process group its watchdog thread polls the events of earlier collectives while this thread records: legal
                # only if the capture's error mode is per thread (kernels launched by the autograd engine's thread into the
                # capturing stream are recorded in either mode)
                capture_begin(g, error_mode=None if self.in_graph_update else "thread_local")
                try:
                    opt.zero_grad()
                    self.result = self.fn()
                    if self.in_graph_update:
                        opt.step(lr, recorded=True)
                finally:
                    g.capture_end()
                self.graph = g
            except CAPTURE_ERRORS as e:
                opt.steps = steps_before
                self.capture_error = e
                torch.cuda.synchronize(dev)
                if strict:
                    raise
                import warnings
                warnings.warn(f"GraphedTrainStep: hipGraph capture failed ({type(e).__name__}: {e}); running eager steps", RuntimeWarning,
                              stacklevel=2)
            finally:
                opt._in_graph = False
        torch.cuda.current_stream(dev).wait_stream(side)
        if self.graph is None and had_hooks:
            opt.overlap_with_backward()
        if self.graph is not None:
            # the capture launched nothing: the step it recorded has not run.  Undo the counter and run it as the first replay would.
            opt.steps = steps_before
        opt._pending = []
        if hasattr(opt, '_left'):
            opt._arm()

    def _eager(self, lr):
        self.opt.zero_grad()
        self.result = self.fn()
        self.opt.all_reduce_grads()
        self.opt.step(lr)
        return self.result

    def __call__(self, lr=None):
        if self.graph is None:
            return self._eager(lr)
        if self.in_graph_update:
            self.opt.stage_hyper(lr)
            self.graph.replay()
        else:
            self.graph.replay()
            self.opt.all_reduce_grads()
            self.opt.step(lr)
        self.replays += 1
        return self.result
