"""Packed convolution weights kept across training steps and refreshed by ONE launch (``sgv3d_gather_pack``).

A training step of the R50 model repacked the current weights of every layer twice per step -- for the forward kernel and, rotated and
transposed, for the data gradient -- in ~270 launches of 5-10 us (``pack_weight_kernel``, ``weight_rot180_transpose_kernel``,
``patch_pack_kernel``, the flips / strided copies of the stride-2 phases): 1.5-2 ms of a 32 ms step for weights that change ONCE per
step.  Every one of those packed forms is a permutation of the parameter with zero padding (and, for the bf16 kernels, a rounding), so a
form is (parameter, packed buffer, index map) and all of them are refreshed by one gather.

How the index maps are made: by running the layer's OWN pack path on an index-valued weight tensor (element i holds i + 1; 0 is what
padding produces) and reading the packed result back as indices -- whatever the pack kernels do (k order, tap flips, channel padding,
fragment orders) is reproduced exactly, and the refreshed buffer is bitwise what the pack kernels would write
(``tests/test_pack_cache_gpu.py``).  bf16 forms carry the index as three base-256 digits (bf16 holds 0..256 exactly).

Validity protocol (a stale packed weight is a silent training bug, so the cache is used only inside a window it controls):
``DataParallelAdamW.zero_grad()`` refreshes every registered form from the parameters as they are NOW and opens the window;
``DataParallelAdamW.step()`` closes it.  Outside the window -- a loop that zeroes gradients some other way, an evaluation forward in
training mode, parameters written between steps through ``.data`` or the flat buckets -- every layer packs per call as before.  Inside
it a parameter modified through torch (``_version``) drops its entry.  Forms that are NOT permutations (Winograd transforms, the
three-plane f32x3 split) make their layer leave the cache (it packs per call again).

Only parameters of the optimiser that opened the window are tracked; they are looked up by address, which the optimiser's flat buckets
keep alive and fixed."""
import os

import numpy as np
import torch

from . import _lib

ENABLED = os.environ.get("SGV3D_PACK_CACHE", "1") != "0"
ACTIVE = None            # the PackCache whose window is open (set by the optimiser's zero_grad, cleared by its step)


class NotAPermutation(Exception):
    """Raised by a PackedConv bound to a cache entry when a kernel asks for a packed form that is not a permutation of the weights."""


class _Job:
    __slots__ = ("dst", "idx", "bf16", "getter")

    def __init__(self, dst, idx, bf16, getter):
        self.dst, self.idx, self.bf16, self.getter = dst, idx, bf16, getter


class _Entry:
    def __init__(self, cache, param, make):
        self.cache, self.param, self.make = cache, param, make
        self.pc = None
        self.version = None
        self.tracked = True
        self.jobs = {}
        self.pc_epoch = -1
        self.unregistered = False    # a permutation form was made during a stream capture: no index map yet

    def _fresh(self):
        return self.make(self.param.detach())

    def conv(self):
        """The layer's PackedConv: the kept one inside the window, a fresh one (packed from the current weights) otherwise."""
        if not (self.tracked and self.cache.open):
            return self._fresh()
        if self.pc is None or self.version != self.param._version:
            if self.jobs:
                self.jobs.clear()
                self.cache.dirty = True
            self.pc = self._fresh()
            self.pc._entry = self
            self.pc_epoch = self.cache.epoch
            self.version = self.param._version
        return self.pc

    def call(self, x, *args, **kw):
        pc = self.conv()
        try:
            return pc(x, *args, **kw)
        except NotAPermutation:
            # the kernel chosen for this launch reads a transformed form: this layer packs per call from now on
            self.tracked = False
            self.pc = None
            if self.jobs:
                self.jobs.clear()
                self.cache.dirty = True
            return self._fresh()(x, *args, **kw)

    def register(self, name, packed, getter):
        """``packed``: the form ``getter(pc)`` just made for this entry's PackedConv.  Derives its index map; from the next
        ``refresh()`` on the buffer is rewritten by the gather launch."""
        if name in self.jobs or not self.tracked:
            return
        # the kept object was built in an earlier step: what it packs from (its own rotated / sliced copy of the weights) is that
        # step's -- the form it just made has to be overwritten from the parameter
        old_source = self.pc_epoch != self.cache.epoch
        if torch.cuda.is_current_stream_capturing():
            if old_source:
                raise NotAPermutation(name + ": first use of this form during a capture, on weights of an earlier step")
            # made by a captured pack launch: that node refreshes it in every replay; an eager step later rebuilds this layer's object
            self.unregistered = True
            return
        p = self.param
        n = p.numel()
        assert n < (1 << 24), "index-valued weights are exact in f32 up to 2^24 elements"
        is_bf16 = packed.dtype in (torch.bfloat16, torch.uint8)
        flat = packed.view(torch.bfloat16).reshape(-1) if packed.dtype == torch.uint8 else packed.reshape(-1)
        v = torch.arange(1, n + 1, dtype=torch.int64, device=p.device)
        if is_bf16:
            digits = []
            for k in range(3):
                src = ((v >> (8 * k)) & 255).to(torch.float32).view(p.shape)
                out = getter(self.make(src))
                out = out.view(torch.bfloat16) if out.dtype == torch.uint8 else out
                digits.append(out.reshape(-1).to(torch.int64))
            got = digits[0] + (digits[1] << 8) + (digits[2] << 16)
        else:
            got = getter(self.make(v.to(torch.float32).view(p.shape))).reshape(-1).to(torch.int64)
        assert got.numel() == flat.numel(), (name, got.numel(), flat.numel())
        idx = (got - 1).to(torch.int32).contiguous()
        job = self.jobs[name] = _Job(flat, idx, is_bf16, getter)
        self.cache.dirty = True
        if old_source:
            self.cache.gather_now([(self.param, job)], p.device)


class PackCache:
    def __init__(self, params):
        self.by_ptr = {p.data_ptr(): p for p in params}
        self.entries = {}
        self.open = False
        self.dirty = False
        self._table = None
        self._tables = []           # every table a captured graph may still read
        self._total_blocks = 0
        self.generation = 0
        self.epoch = 0              # refresh() calls so far: the step whose weights a kept PackedConv was built from

    def lookup(self, weight, key, make):
        """The entry of (parameter, layer geometry), or None when ``weight`` is not one of this optimiser's parameters."""
        p = self.by_ptr.get(weight.data_ptr())
        if p is None or p.shape != weight.shape:
            return None
        k = (weight.data_ptr(), key)
        e = self.entries.get(k)
        if e is None:
            e = self.entries[k] = _Entry(self, p, make)
        return e

    def jobs(self):
        return [j for e in self.entries.values() if e.tracked for j in e.jobs.values()]

    @staticmethod
    def _table_of(pairs, dev):
        """(device table, total blocks) for [(parameter, job)]."""
        lib = _lib.load()
        per, nbytes = lib.sgv3d_gather_pack_elements_per_block(), lib.sgv3d_gather_pack_job_bytes()
        dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("idx", "<u8"), ("n", "<i8"), ("first_block", "<i4"), ("bf16", "<i4")])
        assert dt.itemsize == nbytes, (dt.itemsize, nbytes)
        rows, first = [], 0
        for p, j in pairs:
            n = j.dst.numel()
            rows.append((p.data_ptr(), j.dst.data_ptr(), j.idx.data_ptr(), n, first, 1 if j.bf16 else 0))
            first += (n + per - 1) // per
        return torch.from_numpy(np.array(rows, dtype=dt).view(np.uint8).copy()).to(dev), first

    def gather_now(self, pairs, dev):
        table, blocks = self._table_of(pairs, dev)
        with torch.cuda.device(dev):
            rc = _lib.load().sgv3d_gather_pack(table.data_ptr(), len(pairs), blocks, _lib.stream_handle(dev))
        _lib.check(rc, "sgv3d_gather_pack")
        table.record_stream(torch.cuda.current_stream(dev))

    def _build_table(self, dev):
        pairs = [(e.param, j) for e in self.entries.values() if e.tracked for j in e.jobs.values()]
        keep = [j for _, j in pairs]
        self._n_jobs = len(pairs)
        self._total_blocks = 0
        if pairs:
            self._table, self._total_blocks = self._table_of(pairs, dev)
            # a captured step keeps launching the gather with THIS table: its buffers and index maps stay allocated with it
            self._tables.append((self._table, keep))
        else:
            self._table = None
        self.dirty = False
        self.generation += 1

    def refresh(self, dev):
        """Rewrite every registered packed form from the parameters (one launch) and open the window."""
        if not torch.cuda.is_current_stream_capturing():
            for e in self.entries.values():
                if e.unregistered:
                    e.unregistered, e.pc = False, None
                    if e.jobs:
                        e.jobs.clear()
                        self.dirty = True
        if self.dirty:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.SGV3DError("pack cache: a packed weight form was registered after the warm-up steps; run one more eager "
                                      "step before capturing the training step")
            self._build_table(dev)
        if self._table is not None:
            with torch.cuda.device(dev):
                rc = _lib.load().sgv3d_gather_pack(self._table.data_ptr(), self._n_jobs, self._total_blocks, _lib.stream_handle(dev))
            _lib.check(rc, "sgv3d_gather_pack")
        for e in self.entries.values():
            if e.pc is not None:
                e.version = e.param._version
        self.epoch += 1
        self.open = True

    def close(self):
        self.open = False


def entry_for(weight, key, make):
    """The cache entry for this launch, or None (no open window / not a tracked parameter / switched off)."""
    c = ACTIVE
    if c is None or not c.open or not ENABLED or not weight.is_cuda:
        return None
    return c.lookup(weight, key, make)
