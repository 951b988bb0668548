"""Flat gradient storage + multi-tensor optimizer ops for the GAN trainer.

The reference walks `model.parameters()` (1324 tensors) in Python for every gradient stash, abs-mean, balanced add,
clip, NaN assert and Adam update (trainer/hw_with_style_trainer.py:300-391), each with its own kernel launches and
host syncs. Here every parameter's `.grad` is a persistent view into one flat fp32 buffer, ordered
[main-optimizer params | discriminator-optimizer params | the rest], so that

  * zeroing / stashing a gradient set is one memset / memcpy,
  * per-tensor statistics and updates are single launches of the hwg multi-tensor kernels over static chunk tables,
  * a data-parallel all-reduce is one collective over the flat buffer.

`None` gradients matter in the reference (Adam skips them, stashes record them, the balance ignores them). That state is
tracked on the host in `touched` (set by post-accumulate-grad hooks, cleared by zero_grad) and turned into pointer masks
for the kernels, so the semantics are identical without any device->host traffic.
"""
import math

import os

import numpy as np
import torch

from .. import _lib as L
from .. import ops

CHUNK = 65536
BALANCE_SETS = bool(int(os.environ.get("HWG_BALANCE_SETS", "1") or 1))      # 0: one abs-sum / one add launch per stashed set (A/B, tests)


class FlatParams:
    def __init__(self, params, groups, names=None):
        """params: list of all model parameters (model.parameters() order); groups: dict name -> list of params (disjoint).
        Parameters in no group form the group 'rest'. names (optional, parallel to params): dotted parameter names; their first component
        (the sub-network) defines the `segments` that data-parallel stash reductions travel in."""
        self.params = list(params)
        self.names = list(names) if names is not None else None
        self.device = self.params[0].device
        ids = {id(p): i for i, p in enumerate(self.params)}
        order, self.group_range = [], {}
        seen = set()
        for name, plist in groups.items():
            start = len(order)
            for p in plist:
                order.append(ids[id(p)]); seen.add(ids[id(p)])
            self.group_range[name] = (start, len(order))
        start = len(order)
        order += [i for i in range(len(self.params)) if i not in seen]
        self.group_range["rest"] = (start, len(order))
        for name, (a, b) in self.group_range.items():
            for k in range(a, b):
                self.params[order[k]]._hwg_group = (id(self), name)   # packed-weight cache epoch (ops.WEIGHT_EPOCH)
        self.order = order                      # flat position -> index in self.params
        self.pos = {pi: k for k, pi in enumerate(order)}   # param index -> flat position
        self.nt = len(order)
        numel = np.array([self.params[i].numel() for i in order], dtype=np.int64)
        # 16-byte aligned segments
        padded = (numel + 3) // 4 * 4
        self.offsets = np.concatenate([[0], np.cumsum(padded)[:-1]]).astype(np.int64)
        self.total = int(padded.sum())
        self.numel = numel
        self.flat_grad = torch.zeros(self.total, dtype=torch.float32, device=self.device)
        self.touched = np.zeros(self.nt, dtype=bool)
        for k, pi in enumerate(order):
            p = self.params[pi]
            if not p.requires_grad:
                continue
            p.grad = self.flat_grad[self.offsets[k]: self.offsets[k] + numel[k]].view_as(p)
            p.register_post_accumulate_grad_hook(self._make_hook(k))   # gradients that still arrive through autograd
            p._hwg_touch = self._make_touch(k)                            # gradients the kernels accumulate in place (ops._grad_buffer)
            p._hwg_flat = (self, k)                                       # lets banks of parameters mark many tensors touched at once
        # static tables
        ct, co = [], []
        for k in range(self.nt):
            for off in range(0, int(numel[k]), CHUNK):
                ct.append(k); co.append(off)
        self.nchunks = len(ct)
        self.d_chunk_tensor = torch.tensor(ct, dtype=torch.int32, device=self.device)
        self.d_chunk_off = torch.tensor(co, dtype=torch.int64, device=self.device)
        self.d_numel = torch.from_numpy(numel).to(self.device)
        self._param_ptrs = np.array([self.params[i].data_ptr() for i in order], dtype=np.int64)
        self.d_param_ptrs = torch.from_numpy(self._param_ptrs).to(self.device)
        self._flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._scanned = False
        self._zero_all = bool(int(os.environ.get("HWG_ZERO_ALL", "0") or 0))     # 1: zero_grad fills the whole group slice (the pre-round-6 behaviour; A/B, debugging)
        self._stash_pool = []
        # segments: maximal runs of flat positions that belong to one sub-network (or, without names, one optimizer group)
        label = [(names[pi].split(".")[0] if names is not None else None) for pi in order]
        gname = [None] * self.nt
        for name, (a, b) in self.group_range.items():
            for k in range(a, b):
                gname[k] = name
        self.segment_of = np.zeros(self.nt, dtype=np.int64)
        self.segments = []          # [(first float, one past the last float)]
        for k in range(self.nt):
            if k == 0 or (label[k], gname[k]) != (label[k - 1], gname[k - 1]):
                self.segments.append([int(self.offsets[k]), 0])
            self.segments[-1][1] = int(self.offsets[k] + padded[k])
            self.segment_of[k] = len(self.segments) - 1

    def _make_touch(self, k):
        def touch():
            self.touched[k] = True
        return touch

    def _make_hook(self, k):
        def hook(p):
            # (fires for every parameter whose AccumulateGrad node a backward pass reaches, also when the op accumulated in place through
            # ops._grad_buffer and returned None.) Under a gradient-set redirect the pass belongs to that set: mark its mask, not the current one
            gs = ops.GRAD_SET
            if gs is not None:
                gs[1][k] = True
            else:
                self.touched[k] = True
        return hook

    # -- pointer tables -----------------------------------------------------------------------------
    def base_ptrs(self, flat):
        return flat.data_ptr() + self.offsets * 4

    def masked_ptrs(self, flat, mask):
        return ops.h2d(self.base_ptrs(flat) * mask.astype(np.int64), self.device)

    def group_mask(self, name):
        m = np.zeros(self.nt, dtype=bool)
        a, b = self.group_range[name]
        m[a:b] = True
        return m

    def group_slice(self, flat, name):
        a, b = self.group_range[name]
        if a == b:
            return flat[0:0]
        end = self.offsets[b - 1] + (self.numel[b - 1] + 3) // 4 * 4
        return flat[self.offsets[a]: end]

    def _st(self):
        return ops._stream()

    # -- the reference's gradient bookkeeping, vectorised ------------------------------------------------
    def zero_grad(self, group):
        """optimizer.zero_grad() of torch >= 2.0 (set_to_none): gradients of that optimizer's parameters become None"""
        a, b = self.group_range[group]
        tm = np.zeros(self.nt, dtype=bool)
        tm[a:b] = self.touched[a:b]
        if self.flat_grad.is_cuda and not self._zero_all:
            # only the tensors that HAVE a gradient hold anything but zeros (the buffer starts zeroed, a stash moves-and-zeroes, untouched
            # tensors are never written): one masked multi-tensor pass over those instead of a 143 MB fill per iteration - and nothing at
            # all in the lessons that come right after a stash or that never touched this group
            if tm.any():
                L.call("hwg_mt_unary", self.masked_ptrs(self.flat_grad, tm), None, 0, 0.0, None, self.d_numel, self.d_chunk_tensor, self.d_chunk_off,
                       self.nchunks, CHUNK, self._st())
        else:
            self.group_slice(self.flat_grad, group).zero_()
        self.touched[a:b] = False

    def stash(self):
        """clone every non-None gradient and zero it in place (trainer :305-311, :316-322, :331-338)"""
        ops.join_side_stream()   # weight gradients enqueued on the side stream must have landed before the buffer is read
        if self._stash_pool:
            buf, dirty = self._stash_pool.pop()
        else:
            buf, dirty = torch.zeros_like(self.flat_grad), np.zeros(self.nt, dtype=bool)
        if not self.flat_grad.is_cuda:
            buf.copy_(self.flat_grad)
            self.flat_grad.zero_()
            return (buf, self.touched.copy())
        # One pass over the tensors that HAVE a gradient (buf = grad, grad = 0) instead of a copy and a fill of the whole 190 MB buffer - a gen
        # lesson touches a third of it. A stash must hold zeros where there is no gradient (a tensor another rank or shard touched is summed
        # into it), so the slots an earlier use of `buf` left non-zero (`dirty`, kept with the pooled buffer) and that are not overwritten
        # now are cleared first; the gradient buffer's other slots are zero already (zero_grad / the previous stash left them so).
        tm = self.touched
        stale = dirty & ~tm
        rows = [self.base_ptrs(self.flat_grad) * tm.astype(np.int64), self.base_ptrs(buf) * tm.astype(np.int64)]
        if stale.any():
            rows.append(self.base_ptrs(buf) * stale.astype(np.int64))
        tab = ops.h2d(np.stack(rows), self.device)
        if stale.any():
            L.call("hwg_mt_unary", tab[2], None, 0, 0.0, None, self.d_numel, self.d_chunk_tensor, self.d_chunk_off, self.nchunks, CHUNK, self._st())
        L.call("hwg_mt_unary", tab[0], tab[1], 4, 0.0, None, self.d_numel, self.d_chunk_tensor, self.d_chunk_off, self.nchunks, CHUNK, self._st())
        return (buf, tm.copy())

    def release(self, stash):
        """back to the pool, with the mask of the slots that may be non-zero now (the stash's own mask: an all-reduce widens it to the union)"""
        self._stash_pool.append((stash[0], np.array(stash[1], dtype=bool, copy=True)))

    def abs_sums(self, flat, mask):
        out = torch.empty(self.nt, dtype=torch.float64, device=self.device)
        part = torch.empty(self.nchunks, dtype=torch.float64, device=self.device)
        L.call("hwg_mt_abs_sum", self.masked_ptrs(flat, mask), self.d_numel, self.d_chunk_tensor, self.d_chunk_off, self.nchunks, CHUNK, self.nt, part, out,
               self._st())
        return out

    def balance(self, stashes, multipliers):
        """p.grad += x_k * R_k * mean|p.grad| / mean|R_k| for every stashed set k (trainer :340-377)"""
        if not stashes:
            return
        ops.join_side_stream()
        ns = len(stashes)
        # one pointer table for everything: row 0 the current gradients, rows 1.. the stashed sets (0 = no gradient)
        tab = np.zeros((ns + 1, self.nt), dtype=np.int64)
        tab[0] = self.base_ptrs(self.flat_grad) * self.touched.astype(np.int64)
        for k, st in enumerate(stashes):
            buf, tm = st[0], st[1]
            tab[k + 1] = self.base_ptrs(buf) * tm.astype(np.int64)
            if (tm & ~self.touched).any():
                raise RuntimeError("a stashed gradient exists for a parameter whose current gradient is None (the reference would raise here too)")
        d_tab = ops.h2d(tab, self.device)
        xs = ops.h2d(np.array([float(multipliers[k]) for k in range(ns)], dtype=np.float32), self.device)
        coef = torch.empty((ns, self.nt), dtype=torch.float32, device=self.device)
        if BALANCE_SETS and ns <= 8:
            # mean |.| of the current gradients and of every set in two launches, the balanced adds of all sets in one pass over the gradients
            # (hwg_mt_abs_sum_sets / hwg_mt_axpy_sets: per set / per element the arithmetic of the one-set calls below, bit for bit)
            sums = torch.empty((ns + 1, self.nt), dtype=torch.float64, device=self.device)
            part = torch.empty((ns + 1) * self.nchunks, dtype=torch.float64, device=self.device)
            L.call("hwg_mt_abs_sum_sets", d_tab, ns + 1, self.d_numel, self.d_chunk_tensor, self.d_chunk_off, self.nchunks, CHUNK, self.nt, part, sums,
                   self._st())
            L.call("hwg_mt_balance_coef", sums[0], sums[1:], self.d_numel, d_tab[0], d_tab[1:], xs, ns, self.nt, coef, self._st())
            L.call("hwg_mt_axpy_sets", d_tab[0], d_tab[1:], coef, ns, self.nt, self.d_numel, self.d_chunk_tensor, self.d_chunk_off, self.nchunks, CHUNK,
                   self._st())
            return
        sums = torch.empty((ns + 1, self.nt), dtype=torch.float64, device=self.device)
        part = torch.empty(self.nchunks, dtype=torch.float64, device=self.device)
        for k in range(ns + 1):
            L.call("hwg_mt_abs_sum", d_tab[k], self.d_numel, self.d_chunk_tensor, self.d_chunk_off, self.nchunks, CHUNK, self.nt, part, sums[k], self._st())
        L.call("hwg_mt_balance_coef", sums[0], sums[1:], self.d_numel, d_tab[0], d_tab[1:], xs, ns, self.nt, coef, self._st())
        for k in range(ns):
            L.call("hwg_mt_axpy", d_tab[0], d_tab[k + 1], coef[k], self.d_numel, self.d_chunk_tensor, self.d_chunk_off, self.nchunks, CHUNK, self._st())

    def clip_(self, value):
        """torch.nn.utils.clip_grad_value_ over every parameter that has a gradient"""
        ops.join_side_stream()
        L.call("hwg_mt_unary", self.masked_ptrs(self.flat_grad, self.touched), None, 1, float(value), None, self.d_numel, self.d_chunk_tensor,
               self.d_chunk_off, self.nchunks, CHUNK, self._st())

    def params_nonfinite_flag(self):
        """device flag (int32[1]) set when any parameter holds NaN/inf - the reference asserts per tensor with a host sync each. STICKY: never
        cleared (training aborts on the first non-zero read). The full scan runs once per FlatParams (weights as loaded); afterwards
        HipAdam.step(clip=...) raises the flag where it writes a non-finite parameter value - the only way one can appear."""
        self._scanned = True
        L.call("hwg_mt_unary", self.d_param_ptrs, None, 2, 0.0, self._flag, self.d_numel, self.d_chunk_tensor, self.d_chunk_off, self.nchunks, CHUNK,
               self._st())
        return self._flag


# data-parallel traffic counters (bench.py reports them per step so that a first multi-GPU run is self-diagnosing)
COMM = {"collectives": 0, "bytes": 0}


def _count(t):
    COMM["collectives"] += 1
    COMM["bytes"] += t.numel() * t.element_size()


# HWG_FORCE_DP=1: run the data-parallel exchange even in a one-rank process group (the collectives are identities there). It exists so that
# the RCCL code path - asynchronous whole-buffer reductions under the backward passes, span reductions, the mask exchange - can be
# exercised and timed on a single GPU; results are bit-identical to the plain single-process step.
FORCE_DP = bool(int(os.environ.get("HWG_FORCE_DP", "0") or 0))


_CTL = [None]


def _require_group():
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("data-parallel exchange requested (WORLD_SIZE > 1 or HWG_FORCE_DP=1) but torch.distributed is not initialised: "
                           "launch through torch.distributed.run, or call dist.init_process_group first")


def control_group():
    """gloo group for host-side decisions all ranks take together (which tensors received a gradient somewhere, whether to skip an
    iteration): CPU tensors, so the exchange never waits for a GPU stream. Created collectively on first use."""
    import torch.distributed as dist
    _require_group()
    if _CTL[0] is None:
        # single-node jobs (all ranks local): pin gloo to the loopback interface - its default picks the interface by resolving the host
        # name, which containers do not always provide
        # (only when the launcher says so - LOCAL_WORLD_SIZE == WORLD_SIZE - or the rendezvous address is the loopback; a multi-node
        # srun / mpirun launch that exports RANK / WORLD_SIZE / MASTER_ADDR only must keep gloo's own interface choice)
        lws = os.environ.get("LOCAL_WORLD_SIZE")
        if (lws is not None and lws == os.environ.get("WORLD_SIZE", "1")) or os.environ.get("MASTER_ADDR", "") in ("127.0.0.1", "localhost", "::1"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        _CTL[0] = dist.new_group(backend="gloo")
    return _CTL[0]


def _span_cache(flat):
    """per FlatParams instance: key (lesson, stash position) -> sorted list of segment indices every rank reduces for that key. (It lived in a
    module-level dict keyed by id(flat) once: a FlatParams built later in the same process could inherit the id - and the segment indices -
    of a freed one.)"""
    c = getattr(flat, "_span_cache", None)
    if c is None:
        c = flat._span_cache = {}
    return c


def segment_spans(flat, segs):
    """float ranges of a set of segments, adjacent ones merged"""
    spans = []
    for i in sorted(segs):
        a, b = flat.segments[i]
        if spans and spans[-1][1] == a:
            spans[-1][1] = b
        else:
            spans.append([a, b])
    return [(a, b) for a, b in spans]


def start_stash_allreduce(stash, world, flat=None, key=None):
    """Data parallel: begin the SUM all-reduce of a freshly stashed gradient set without waiting for it. The collectives run on the
    communicator's stream behind the stash copy, so they overlap the backward passes that follow (an `auto` lesson stashes four sets
    before it balances them). Every rank must reduce the SAME ranges, and which tensors a set touches differs between ranks (a character
    expert is touched only where that character occurs), so the ranges are whole sub-network segments, agreed once: the first time a
    (lesson, stash position) `key` occurs the None-masks are OR-ed over the control group right here (a blocking host exchange) and the
    segments the OR-ed mask touches are remembered; every later occurrence starts its reductions at once from the remembered segments, and
    the masks travel with the ONE exchange `allreduce_gradient_sets` makes per lesson anyway (which also verifies the remembered
    segments still cover the set, and widens them - on every rank alike - if they do not). Without `flat` the whole buffer travels.
    Returns the stash as a list [buffer, mask, pending work(s), mask already exchanged, segments reduced, key]."""
    import torch.distributed as dist
    st = [stash[0], stash[1], None, False, None, key]
    if world > 1 or FORCE_DP:
        _require_group()
        if flat is None:
            _count(st[0])
            st[2] = [(dist.all_reduce(st[0], op=dist.ReduceOp.SUM, async_op=True), st[0])]
        else:
            segs = _span_cache(flat).get(key) if key is not None else None
            if segs is None:
                m = torch.from_numpy(np.asarray(st[1]).astype(np.int32))
                _count(m)
                dist.all_reduce(m, op=dist.ReduceOp.MAX, group=control_group())
                st[1][:] = m.numpy().astype(bool)
                st[3] = True
                segs = sorted(set(flat.segment_of[np.nonzero(st[1])[0]].tolist()))
                if key is not None:
                    _span_cache(flat)[key] = segs
            st[4] = list(segs)
            st[2] = []
            for a, b in segment_spans(flat, segs):
                view = st[0][a:b]
                _count(view)
                st[2].append((dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True), view))
    return st


def touched_spans(flat, mask, max_spans=8, min_gap=1 << 16):
    """[(start, end)] float ranges of the flat buffer that cover every touched tensor: runs of touched tensors, merged across gaps
    shorter than `min_gap` floats and then across the smallest gaps until at most `max_spans` remain. The mask is the OR over all
    ranks, so every rank derives the same ranges; everything outside them is zero on every rank and needs no exchange (a `disc`
    lesson touches 6 MB of the 190 MB buffer)."""
    idx = np.nonzero(mask)[0]
    if idx.size == 0:
        return []
    starts = flat.offsets[idx]
    ends = starts + (flat.numel[idx] + 3) // 4 * 4
    spans = [[int(starts[0]), int(ends[0])]]
    for a, b in zip(starts[1:], ends[1:]):
        if a - spans[-1][1] < min_gap:
            spans[-1][1] = int(b)
        else:
            spans.append([int(a), int(b)])
    while len(spans) > max_spans:
        gaps = [spans[i + 1][0] - spans[i][1] for i in range(len(spans) - 1)]
        i = int(np.argmin(gaps))
        spans[i][1] = spans[i + 1][1]
        del spans[i + 1]
    return [(a, b) for a, b in spans]


def allreduce_gradient_sets(flat, stashes, world, device):
    """Data-parallel averaging of the current gradient set and of every stashed set.
    The None-masks are OR-ed first (int32 MAX): a tensor that received a gradient on any rank exists, possibly as zeros, on all.
    Stashes whose reduction was started early (start_stash_allreduce, whole buffer, hidden under the later backward passes) are only
    waited for; everything reduced here - the current set, which sits on the critical path - is exchanged only over the ranges that
    hold touched tensors."""
    import torch.distributed as dist
    if world == 1 and not FORCE_DP:
        return
    _require_group()
    ops.join_side_stream()
    # the None-masks are host state (set while the backward pass is being enqueued), so they are OR-ed over the gloo control group: no
    # device collective + read-back, the host keeps its run-ahead over the GPU (a device MAX all-reduce here cost 4 % of the step)
    todo = [k for k, s in enumerate(stashes) if not (len(s) > 3 and s[3])]       # stashes whose mask has not been exchanged yet
    masks = [flat.touched] + [stashes[k][1] for k in todo]
    m = torch.from_numpy(np.stack(masks).astype(np.int32))
    _count(m)
    dist.all_reduce(m, op=dist.ReduceOp.MAX, group=control_group())
    m = m.numpy().astype(bool)
    flat.touched[:] = m[0]
    for i, k in enumerate(todo):
        stashes[k][1][:] = m[1 + i]
    pending = []
    for k, s in enumerate(stashes):
        works = s[2] if len(s) > 2 else None
        if works is not None:
            pending.extend(works)
            if len(s) > 4 and s[4] is not None:
                # started from remembered segments: the OR-ed mask (identical on every rank) must lie inside them; a tensor outside means
                # some rank touched a sub-network this (lesson, stash position) never touched before - reduce those segments now, on every
                # rank alike, and remember them
                need = set(flat.segment_of[np.nonzero(s[1])[0]].tolist()) - set(s[4])
                if need:
                    for a, b in segment_spans(flat, need):
                        view = s[0][a:b]
                        _count(view)
                        pending.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True), view))
                    s[4] = sorted(set(s[4]) | need)
                    if len(s) > 5 and s[5] is not None:          # only the entry of this stash's own key is widened
                        _span_cache(flat)[s[5]] = list(s[4])
        else:
            for a, b in touched_spans(flat, s[1]):
                view = s[0][a:b]
                _count(view)
                pending.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True), view))
    for a, b in touched_spans(flat, m[0]):
        view = flat.flat_grad[a:b]
        _count(view)
        pending.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, async_op=True), view))
    for work, buf in pending:
        work.wait()
        buf.div_(world)
    for s in stashes:
        if len(s) > 2:
            s[2] = None


class HipAdam:
    """torch.optim.Adam (betas, eps, no weight decay / amsgrad) over one parameter group of a FlatParams, as a single
    multi-tensor kernel launch. Tensors whose gradient is None are skipped and their step count does not advance."""

    def __init__(self, flat, group, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, **unused):
        if weight_decay:
            raise NotImplementedError("weight decay is 0 in every shipped config")
        self.flat, self.group = flat, group
        self.lr, self.betas, self.eps = lr, tuple(betas), eps
        self.exp_avg = torch.zeros_like(flat.flat_grad)
        self.exp_avg_sq = torch.zeros_like(flat.flat_grad)
        self.steps = np.zeros(flat.nt, dtype=np.int64)
        self.mask = flat.group_mask(group)
        self.param_groups = [{"lr": lr, "betas": self.betas, "eps": eps, "params": [flat.params[flat.order[k]] for k in np.nonzero(self.mask)[0]]}]

    def zero_grad(self):
        self.flat.zero_grad(self.group)

    def step(self, clip=None):
        """clip = None: optimizer.step() on the gradients as they are. clip = v: the trainer's whole gradient-consumption tail in ONE launch
        (hwg_mt_clip_adam): clip_grad_value_(v) over EVERY touched tensor (also those of the optimizer that does not step now), this
        optimizer's Adam update, and the non-finite check of the parameters it writes (FlatParams._flag, sticky) - it replaces a clip pass
        over the gradients, a scan over all 47 M parameters and the Adam launch (and their uploads) per stepping lesson."""
        f = self.flat
        active = self.mask & f.touched
        if not active.any() and clip is None:
            return
        ops.join_side_stream()
        self.steps[active] += 1
        lr = self.param_groups[0]["lr"]
        b1, b2 = self.betas
        t = np.maximum(self.steps, 1).astype(np.float64)
        step_size = (lr / (1.0 - b1 ** t)).astype(np.float32)
        bc2 = np.sqrt(1.0 - b2 ** t).astype(np.float32)
        am = active.astype(np.int64)
        gm = am if clip is None else f.touched.astype(np.int64)
        nt = f.nt
        # ONE upload: the four pointer rows (int64) followed by the two per-tensor scalars (float32)
        host = np.empty(4 * nt * 8 + 2 * nt * 4, dtype=np.uint8)
        host[:4 * nt * 8].view(np.int64).reshape(4, nt)[:] = np.stack([f._param_ptrs * am, f.base_ptrs(f.flat_grad) * gm, f.base_ptrs(self.exp_avg) * am,
                                                                      f.base_ptrs(self.exp_avg_sq) * am])
        host[4 * nt * 8:].view(np.float32).reshape(2, nt)[:] = np.stack([step_size, bc2])
        dev = ops.h2d(host, f.device)
        d_tab = dev[:4 * nt * 8].view(torch.int64).view(4, nt)
        d_sc = dev[4 * nt * 8:].view(torch.float32).view(2, nt)
        if clip is None:
            L.call("hwg_mt_adam", d_tab[0], d_tab[1], d_tab[2], d_tab[3], d_sc[0], d_sc[1], float(b1), float(b2), float(self.eps), 0.0, f.d_numel,
                   f.d_chunk_tensor, f.d_chunk_off, f.nchunks, CHUNK, f._st())
        else:
            L.call("hwg_mt_clip_adam", d_tab[0], d_tab[1], d_tab[2], d_tab[3], d_sc[0], d_sc[1], float(b1), float(b2), float(self.eps), float(clip),
                   f._flag, f.d_numel, f.d_chunk_tensor, f.d_chunk_off, f.nchunks, CHUNK, f._st())
        if active.any():
            ops.bump_weight_epoch((id(f), self.group))   # parameters changed behind torch's version counters
            ops.repack_group((id(f), self.group))        # ... and their cached packed images are refreshed in one launch

    # checkpoint format of torch.optim.Adam (state keyed by parameter index within the group)
    def state_dict(self):
        f = self.flat
        state = {}
        for j, k in enumerate(np.nonzero(self.mask)[0]):
            if self.steps[k] == 0:
                continue
            p = f.params[f.order[k]]
            sl = slice(int(f.offsets[k]), int(f.offsets[k] + f.numel[k]))
            state[j] = {"step": torch.tensor(float(self.steps[k])), "exp_avg": self.exp_avg[sl].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[sl].view_as(p).clone()}
        n = int(self.mask.sum())
        lr = self.param_groups[0]["lr"]
        groups = [{"lr": lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False, "params": list(range(n))}]
        if self.group == "main":
            # the reference builds its main Adam with a second ("slow", lr x 0.1) group that the shipped configs leave empty
            # (base/base_trainer.py:95-97); torch's load_state_dict insists on the same number of groups, so it is written here too
            groups.append({"lr": 0.1 * lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False, "params": []})
        return {"state": state, "param_groups": groups}

    def reset_state(self):
        """forget every moment and step count (what a freshly constructed torch.optim.Adam holds)"""
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.steps[:] = 0

    def load_state_dict(self, sd):
        f = self.flat
        self.reset_state()        # tensors absent from the loaded state have no state, as in torch.optim.Adam.load_state_dict
        ks = np.nonzero(self.mask)[0]
        for j, st in sd["state"].items():
            k = ks[int(j)]
            sl = slice(int(f.offsets[k]), int(f.offsets[k] + f.numel[k]))
            self.exp_avg[sl].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[sl].copy_(st["exp_avg_sq"].reshape(-1))
            self.steps[k] = int(float(st["step"]))
        if sd.get("param_groups"):
            self.param_groups[0]["lr"] = sd["param_groups"][0].get("lr", self.param_groups[0]["lr"])
