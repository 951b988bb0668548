"""Launch-list replay for the sequential sub-networks (round 5; SURVEY section 7 step 8 without hipGraph's fixed addresses).

The host side of a network pass is one Python -> torch -> C-ABI round trip per kernel: ~10 us of interpreter, allocator and autograd work
on top of the ~7 us the HIP launch itself costs. For a sub-network whose op sequence is a pure function of its input geometry - the frozen
CNN recogniser (model/cnn_only_hwr.py; reference: model/cnn_only_hwr.py:7-107): no random draws, static weights, 45 launches forward and
~70 backward, 7 + 11 passes per curriculum cycle - the sequence is recorded ONCE per (geometry, gradient requirements, trainer switches)
and replayed with one call into the C side of the binding (`_hwgcall.replay`, generated from include/hwg.h like the call thunks).

Recording (second sighting of a key, so that packed-weight and plan caches are warm): a separate forward + backward of the network on a
copy of the input, parameter gradients redirected into a scratch gradient set, BatchNorm running statistics put back afterwards. While it
runs, `_lib.call` logs every entry-point call. Every pointer argument is classified:
  literal        parameters, buffers, cached weight images, descriptors (static addresses; the program keeps the tensors alive)
  slot + offset  intermediates (one region of a per-pass arena per tensor storage, no reuse inside a pass; the forward arena lives in the
                 autograd context like the saved activations it replaces), the pass's inputs (image / upstream gradient), streams, scratch
                 workspaces, and - resolved per replay because they move - the parameters' gradient buffers (`ops._grad_buffer`: honours
                 gradient-set redirects and marks `touched`) and the slices of the deferred-sum arena (`ops._defer_workspace`)
A recorded program is trusted only after a SELF-CHECK: it is replayed on the recording's own input and must reproduce output, input
gradient and every parameter gradient of the eager recording BIT FOR BIT; otherwise (an op that launches outside the C-ABI, an
unclassifiable pointer) the key stays on the eager path for the rest of the process. Replay never re-executes the process and never touches
the oracle; with it disabled (HWG_REPLAY=0, the default until a trainer switches it on) nothing here runs.
"""
import os
import re

import numpy as np
import torch

from . import _lib as L
from . import ops

ENABLED = bool(int(os.environ.get("HWG_REPLAY", "0") or 0))      # bench.py / train.py switch it on (enable()); library default off


def enable(default=True):
    """switch the replay on unless the environment says otherwise (HWG_REPLAY=0 / 1)"""
    global ENABLED
    ENABLED = bool(int(os.environ.get("HWG_REPLAY", "1" if default else "0") or 0))
    return ENABLED
# Sightings of a geometry before it is recorded. A recording costs ~8 extra passes of the network (one eager backward and one replayed
# self-check per backward variant): worth it for geometries that keep coming back - the recogniser sees the REAL lines at the batch's padded
# width again and again (no input gradient: 3 sightings) - and a loss for the long tail of widths the generated lines take (input gradient
# wanted: 24 sightings, so that only the handful of most frequent widths is ever recorded). No eviction: once MAX_PROGRAMS geometries are
# recorded the rest stays eager (evicting under a spread of widths would keep re-recording).
RECORD_AFTER = int(os.environ.get("HWG_REPLAY_AFTER", "3") or 3)
RECORD_AFTER_DX = int(os.environ.get("HWG_REPLAY_AFTER_DX", "24") or 24)
MAX_PROGRAMS = int(os.environ.get("HWG_REPLAY_PROGRAMS", "12") or 12)   # recorded geometries kept (each owns arenas of a few hundred MB)
STATS = {"captures": 0, "rejected": 0, "fwd": 0, "bwd": 0, "eager": 0}
_FUNCS = L._hwgcall.replay_functions() if getattr(L, "_hwgcall", None) is not None and hasattr(L._hwgcall, "replay_functions") else {}

# slots of the pointer table that every program has; gradient buffers and deferred-sum slices follow
S_FWD, S_BWD, S_X, S_DY, S_STREAM, S_SIDE, S_WS, S_WS_SIDE, S_FIRST = 0, 1, 2, 3, 4, 5, 6, 7, 8


class _Reject(Exception):
    pass


class _Phase:
    """one recorded call list (forward or backward) in the flat form `_hwgcall.replay` takes"""

    def __init__(self):
        self.recs, self.kinds, self.vals, self.offs = [], [], [], []
        self.bytes = 0            # size of this phase's arena
        self.ws = self.ws_side = 0
        self.grad_params = []     # parameters whose gradient buffer the phase writes (table slots S_FIRST + i)
        self.defer = []           # byte sizes of the deferred-sum slices it asks for (table slots after the gradient buffers)
        self.uses_side = False

    def freeze(self):
        self.recs = np.ascontiguousarray(np.array(self.recs, dtype=np.int32).reshape(-1, 3))
        self.kinds = np.ascontiguousarray(np.array(self.kinds, dtype=np.uint8))
        self.vals = np.ascontiguousarray(np.array(self.vals, dtype=np.int64))
        self.offs = np.ascontiguousarray(np.array(self.offs, dtype=np.int64))
        self.nslots = S_FIRST + len(self.grad_params) + len(self.defer)


class Program:
    def __init__(self):
        self.fwd = _Phase()
        self.bwds = {}            # (DEFER_REDUCE, SIDE_WGRAD) at backward time -> the backward recorded under those switches (the trainers flip
                                  # them around their backward passes; (False, False) is always there and valid under any setting)
        self.keep = []            # static tensors the literal pointers point into
        self.out = None           # (offset in the forward arena, shape) of the network's output
        self.dx = {}              # per backward variant: (offset in its arena, shape) of the input gradient
        self.bn = []              # BatchNorm modules whose host-side batch counter a forward advances
        self.pools = {}           # (phase, stream) -> arenas not in use
        self.verdict = None       # device scalar of the self-check until it has been read (then True)


class _Recorder:
    """replaces `_lib.call` (and the ops helpers that hand out moving buffers) while a recording runs"""

    def __init__(self, prog, static_ptrs, params, device):
        self.prog, self.static, self.params, self.device = prog, static_ptrs, params, device
        self.phase = None                   # None | "fwd" | ("bwd", flags)
        self.hold = []                      # every tensor seen: no storage is recycled while the recording runs
        self.slots = {}                     # storage address -> (phase name, offset in that phase's arena)
        self.dynamic = {}                   # data_ptr of a tensor handed out by a helper -> (table slot, )
        self.inputs = {}                    # storage address -> table slot (S_X / S_DY)
        self.pidx = {id(p): i for i, p in enumerate(params)}
        self.host_static = ops.static_host_ptrs()
        self.main_stream = ops._stream()
        self.side_stream = ops._side_stream(device)[1]      # (created here if no pass has forked yet: the recording's backward variants may)

    def _ph(self):
        if isinstance(self.phase, tuple):
            return self.prog.bwds.setdefault(self.phase[1], _Phase())
        return self.prog.fwd

    # ---- helpers of ops that hand out buffers whose address changes from pass to pass -------------------------------------------------
    def grad_buffer(self, p):
        g = self.orig["_grad_buffer"](p)
        ph = self._ph()
        if id(p) not in self.pidx:
            raise _Reject("gradient buffer of a parameter outside the network")
        ids = [id(q) for q in ph.grad_params]
        if id(p) not in ids:
            ph.grad_params.append(p)
            ids.append(id(p))
        self.dynamic[g.data_ptr()] = ("grad", ids.index(id(p)))
        self.hold.append(g)
        return g

    def defer_workspace(self, nbytes, device):
        t = self.orig["_defer_workspace"](nbytes, device)
        if t is None:
            raise _Reject("deferred-sum arena full while recording")
        ph = self._ph()
        ph.defer.append(int(nbytes))
        self.dynamic[t.data_ptr()] = ("defer", len(ph.defer) - 1)
        self.hold.append(t)
        return t

    def workspace(self, nbytes, device):
        t = self.orig["workspace"](nbytes, device)
        ph = self._ph()
        ph.ws = max(ph.ws, int(nbytes), int(t.numel()))       # (ops pass the buffer's size along: a replay must find at least that much)
        self.dynamic[t.data_ptr()] = ("ws", 0)
        return t

    def side_workspace(self, nbytes, device, stream):
        t = self.orig["_side_workspace"](nbytes, device, stream)
        ph = self._ph()
        ph.ws_side = max(ph.ws_side, int(nbytes), int(t.numel()))
        self.dynamic[t.data_ptr()] = ("ws_side", 0)
        return t

    # ---- the call log -------------------------------------------------------------------------------------------------------------------
    def call(self, name, *args):
        if self.phase is not None:
            self._log(name, args)
        return self.orig_call(name, *args)

    def _log(self, name, args):
        info = _FUNCS.get(name)
        if info is None:
            raise _Reject("%s cannot be replayed" % name)
        fid, kinds = info
        if len(kinds) != len(args):
            raise _Reject("%s: %d arguments recorded, %d declared" % (name, len(args), len(kinds)))
        ph = self._ph()
        ph.recs.append((fid, len(ph.kinds), len(args)))
        for k, a in zip(kinds, args):
            off = 0
            if k == "i":
                kind, val = 0, int(a)
            elif k == "f":
                kind, val = 1, int(np.array([float(a)], dtype=np.float64).view(np.int64)[0])
            elif k == "s":                         # a parameter the header declares as a stream: re-pointed at the streams of the replaying pass
                a = 0 if a is None else int(a)
                if a == self.main_stream:
                    kind, val = 3, S_STREAM
                elif self.side_stream is not None and a == self.side_stream:
                    kind, val = 3, S_SIDE
                    ph.uses_side = True
                else:
                    raise _Reject("%s runs on a stream that is neither the pass's main nor its side stream" % name)
            elif a is None:
                kind, val = 2, 0
            elif isinstance(a, torch.Tensor):
                kind, val, off = self._tensor(a, ph)
            else:
                # a host-side address: only the geometry descriptors of the plan caches have static lifetime. Anything else (the address of a
                # temporary numpy array handed to an out-parameter or a host table) would dangle on replay
                kind, val = 2, int(a)
                if val != 0 and val not in self.host_static:
                    self.host_static = ops.static_host_ptrs()       # (plans made during this recording)
                    if val not in self.host_static:
                        raise _Reject("%s takes a host pointer that is not a cached descriptor" % name)
            ph.kinds.append(kind); ph.vals.append(val); ph.offs.append(off)

    def _tensor(self, t, ph):
        self.hold.append(t)
        dyn = self.dynamic.get(t.data_ptr())
        if dyn is not None:
            what, i = dyn
            if what == "grad":
                return 3, S_FIRST + i, 0
            if what == "defer":
                return 3, -(i + 1), 0               # patched to its slot once the number of gradient buffers is known (_finish)
            return 3, (S_WS if what == "ws" else S_WS_SIDE), 0
        st = t.untyped_storage()
        base = st.data_ptr()
        off = t.data_ptr() - base
        if base in self.static:
            return 2, t.data_ptr(), 0
        slot = self.inputs.get(base)
        if slot is not None:
            return 3, slot, off
        hit = self.slots.get(base)
        if hit is None:
            hit = self.slots[base] = (self.phase, ph.bytes)
            ph.bytes += (st.nbytes() + 255) & ~255
        if hit[0] == "fwd":
            return 3, S_FWD, hit[1] + off
        if hit[0] != self.phase:
            raise _Reject("a pass reads an intermediate of another pass")
        return 3, S_BWD, hit[1] + off

    def _finish(self):
        for ph in [self.prog.fwd] + list(self.prog.bwds.values()):
            ng = len(ph.grad_params)
            ph.vals = [S_FIRST + ng + (-v - 1) if (k == 3 and v < 0) else v for k, v in zip(ph.kinds, ph.vals)]
            ph.freeze()

    # ---- install / remove -----------------------------------------------------------------------------------------------------------------
    def __enter__(self):
        self.orig_call = L.call
        self.orig = {n: getattr(ops, n) for n in ("_grad_buffer", "_defer_workspace", "workspace", "_side_workspace")}
        L.call = self.call
        ops._grad_buffer, ops._defer_workspace, ops.workspace, ops._side_workspace = self.grad_buffer, self.defer_workspace, self.workspace, self.side_workspace
        return self

    def __exit__(self, *exc):
        L.call = self.orig_call
        for n, f in self.orig.items():
            setattr(ops, n, f)
        return False


class _Lease:
    """an arena on loan from its program's pool: fresh blocks of a few hundred MB per pass (sizes differ per geometry) splinter torch's
    allocator cache and end in hipMalloc / hipFree round trips; the pool hands the same block to the next pass on the SAME stream (stream
    order makes that safe: the next user's kernels queue behind the last reader's)"""

    __slots__ = ("pool", "arena")

    def __init__(self, pool, nbytes, device):
        self.pool = pool
        self.arena = pool.pop() if pool else torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)

    def __del__(self):
        self.pool.append(self.arena)


def _lease(prog, which, nbytes, device):
    return _Lease(prog.pools.setdefault((which, ops._stream()), []), nbytes, device)


def _view(arena, off, shape):
    n = 4
    for s in shape:
        n *= int(s)
    return arena[off: off + n].view(torch.float32).view(shape)


def _table(ph, prog, device, fwd_arena, bwd_arena, x, dy):
    """the pointer table of one replay: arenas, inputs, streams, workspaces, then the buffers that move from pass to pass"""
    t = np.zeros(ph.nslots, dtype=np.uint64)
    t[S_FWD] = fwd_arena.data_ptr() if fwd_arena is not None else 0
    t[S_BWD] = bwd_arena.data_ptr() if bwd_arena is not None else 0
    t[S_X] = x.data_ptr()
    t[S_DY] = dy.data_ptr() if dy is not None else 0
    t[S_STREAM] = ops._stream()
    hold = []
    if ph.uses_side:
        s2, raw2, skey = ops._side_stream(device)
        t[S_SIDE] = raw2
        if ph.ws_side:
            t[S_WS_SIDE] = ops._side_workspace(ph.ws_side, device, s2).data_ptr()
        ops._side_dirty.add(skey)
    if ph.ws:
        t[S_WS] = ops.workspace(ph.ws, device).data_ptr()
    for i, p in enumerate(ph.grad_params):
        t[S_FIRST + i] = ops._grad_buffer(p).data_ptr()
    base = S_FIRST + len(ph.grad_params)
    for i, nbytes in enumerate(ph.defer):
        sl = ops._defer_workspace(nbytes, device)
        if sl is None:
            # arena full: a private buffer that lives until the flush does the same job (the eager path sums at once instead; the queued sum
            # of the recording needs the partial images to stay where they are until join_side_stream())
            sl = torch.empty(nbytes, dtype=torch.uint8, device=device)
            ops._defer["count"] += 1
        hold.append(sl)
        t[base + i] = sl.data_ptr()
    return t, hold


def _run(ph, table):
    rc, bad = L._hwgcall.replay(ph.recs, ph.kinds, ph.vals, ph.offs, table)
    if rc != 0:
        raise L.HwgError("replayed call %d failed (%d): %s" % (bad, rc, L.last_error()))


class _ReplayNet(ops.Function):
    """one autograd / tape node for a whole network pass. The parameters travel as inputs for the graph's sake only: their gradients are
    accumulated by the kernels (directly, or into the current gradient set), as on the eager path."""

    @staticmethod
    def forward(ctx, x, prog, *params):
        x = x.contiguous()
        lease = _lease(prog, "f", prog.fwd.bytes, x.device)
        arena = lease.arena
        table, hold = _table(prog.fwd, prog, x.device, arena, None, x, None)
        _run(prog.fwd, table)
        for m in prog.bn:
            m._tracked_pending = getattr(m, "_tracked_pending", 0) + 1
        ctx.prog, ctx.lease, ctx.x = prog, lease, x
        STATS["fwd"] += 1
        # the output leaves the arena (it outlives the pass: losses, the style extractor and the alignment read it later)
        return _view(arena, *prog.out).clone()

    @staticmethod
    def backward(ctx, dy):
        prog = ctx.prog
        fl = (bool(ops.DEFER_REDUCE), bool(ops.SIDE_WGRAD))
        if fl not in prog.bwds:
            fl = (False, False)
        ph = prog.bwds[fl]
        dy = dy.contiguous()
        lease = _lease(prog, "b%d%d" % fl, ph.bytes, dy.device)
        arena = lease.arena
        table, hold = _table(ph, prog, dy.device, ctx.lease.arena, arena, ctx.x, dy)
        _run(ph, table)
        # side-stream weight gradients and queued partial-image sums read these until join_side_stream()
        ops._side_hold.append((ctx.lease, lease, ctx.x, dy, hold))
        STATS["bwd"] += 1
        dxs = prog.dx.get(fl)
        dx = _view(arena, *dxs).clone() if (dxs is not None and ctx.needs_input_grad[0]) else None
        return (dx, None) + (None,) * (len(ctx.needs_input_grad) - 2)


# ---- recording and dispatch ---------------------------------------------------------------------------------------------------------------
_programs = {}      # key -> sightings so far | Program | None (rejected: eager for the rest of the process)
BACKWARD_FLAGS = set()   # (DEFER_REDUCE, SIDE_WGRAD) settings the owner of the network runs backward passes under (the trainers say; else the current ones)


def reset():
    _programs.clear()


def _needs_input_grad(x):
    if ops.TAPE is not None:
        return id(x) in ops.TAPE.live or (x.requires_grad and x.is_leaf)
    return torch.is_grad_enabled() and x.requires_grad


def _static_ptrs(net):
    ptrs, keep = set(), []
    for t in list(net.parameters()) + list(net.buffers()):
        ptrs.add(t.untyped_storage().data_ptr()); keep.append(t)
    mine = {id(p) for p in net.parameters()}
    for e in ops._pack_cache.values():
        if id(e[3]) in mine:
            ptrs.add(e[2].untyped_storage().data_ptr()); keep.append(e[2])
    return ptrs, keep


def _record(net, fn, scope, x, needs_dx, params):
    """-> Program, or raises _Reject. Leaves no trace: gradients go to scratch sets, BatchNorm statistics, counters and the trainer's switches
    are put back."""
    flat = params[0]._hwg_flat[0]
    dev = x.device
    bns = [m for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    saved_bn = [(m.running_mean.clone(), m.running_var.clone(), getattr(m, "_tracked_pending", 0)) for m in bns]

    def restore_bn():
        for m, (rm, rv, pend) in zip(bns, saved_bn):
            m.running_mean.copy_(rm); m.running_var.copy_(rv); m._tracked_pending = pend
    total = int(flat.flat_grad.numel())
    ks = [p._hwg_flat[1] for p in params]
    lo = min(int(flat.offsets[k]) for k in ks)
    hi = max(int(flat.offsets[k]) + int(flat.numel[k]) for k in ks)
    variants = [(False, False)] + sorted(fl for fl in (BACKWARD_FLAGS or {(bool(ops.DEFER_REDUCE), bool(ops.SIDE_WGRAD))}) if fl != (False, False))
    prog = Program()
    prog.bn = [m for m in bns if m.training]
    static, prog.keep = _static_ptrs(net)
    tape, ops.TAPE = ops.TAPE, None
    flags = (ops.DEFER_REDUCE, ops.SIDE_WGRAD)
    scratches = []

    def backward_under(fl, y, g, xc, rec=None):
        """one backward pass under the switches `fl` into a scratch gradient set of its own -> (input gradient, the network's span of the set, mask).
        The join behind the pass (side-stream join + the flush of the queued partial-image sums) is NOT part of the program: `rec` stops logging
        before it. A replayed pass queues its sums like an eager one and the owner's join_side_stream() flushes them (what `_table` / `_side_hold`
        are written for); a flush baked into the call list would also carry the address of the flush's host-side out-parameter."""
        scratch = torch.empty(total, dtype=torch.float32, device=dev)      # (only the network's own span is written, zeroed and compared)
        scratch[lo:hi].zero_()
        scratches.append(scratch)
        mask = np.zeros(len(flat.numel), dtype=bool)
        ops.DEFER_REDUCE, ops.SIDE_WGRAD = fl
        xc.grad = None
        with ops.grad_set((scratch, mask)):
            if y.requires_grad:
                torch.autograd.backward(y, g, retain_graph=True)
            if rec is not None:
                rec.phase = None
            ops.join_side_stream()
        return xc.grad, scratch[lo:hi], mask

    try:
        with torch.enable_grad():
            # ---- the eager recording: one forward, one backward per variant --------------------------------------------------------------
            xc = x.detach().clone().requires_grad_(bool(needs_dx))
            base = {}
            with _Recorder(prog, static, params, dev) as rec:
                rec.inputs[xc.untyped_storage().data_ptr()] = S_X
                rec.phase = "fwd"
                with ops.scope(scope):
                    y = fn(xc)
                rec.phase = None
                # the upstream gradient of the recording: uploaded by a stream-ordered copy (not a C-ABI call, so not part of the program)
                g = torch.randn(y.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float32).pin_memory().to(dev, non_blocking=True)
                rec.inputs[g.untyped_storage().data_ptr()] = S_DY
                hit = rec.slots.get(y.untyped_storage().data_ptr())
                if hit is None or hit[0] != "fwd" or not y.is_contiguous():
                    raise _Reject("the output is not a forward intermediate")
                prog.out = (hit[1] + y.data_ptr() - y.untyped_storage().data_ptr(), tuple(y.shape))
                for fl in variants:
                    rec.phase = ("bwd", fl)
                    rec._ph()
                    dxe, span, mask = backward_under(fl, y, g, xc, rec)
                    rec.phase = None
                    if needs_dx:
                        hit = rec.slots.get(dxe.untyped_storage().data_ptr()) if dxe is not None else None
                        if hit is None or hit[0] != ("bwd", fl) or not dxe.is_contiguous():
                            raise _Reject("the input gradient is not a backward intermediate")
                        prog.dx[fl] = (hit[1] + dxe.data_ptr() - dxe.untyped_storage().data_ptr(), tuple(dxe.shape))
                    base[fl] = (None if dxe is None else dxe.detach().clone(), span, mask)
                rec._finish()
            y0 = y.detach().clone()
            restore_bn()
            # ---- the self-check: the same input through the programs ------------------------------------------------------------------
            # The recording's tensors are released first (a literal pointer into a recorded intermediate must not find its old bytes) and the
            # replay runs in arenas filled with NaN bit patterns: a lane the program never writes (a torch-side fill that was not recorded)
            # then shows up in the comparison instead of hiding behind a zero weight.
            del y, dxe, span
            rec.hold.clear(); rec.slots.clear(); rec.dynamic.clear()
            st_key = ops._stream()
            prog.pools.setdefault(("f", st_key), []).append(torch.full((max(prog.fwd.bytes, 256),), 0xFF, dtype=torch.uint8, device=dev))
            for fl in variants:
                prog.pools.setdefault(("b%d%d" % fl, st_key), []).append(torch.full((max(prog.bwds[fl].bytes, 256),), 0xFF, dtype=torch.uint8, device=dev))
            xr = x.detach().clone().requires_grad_(bool(needs_dx))
            y1 = _ReplayNet.apply(xr, prog, *params)
            ok = (y0 == y1).all()
            for fl in variants:
                dxr, span, mask = backward_under(fl, y1, g, xr)
                dx0, span0, mask0 = base[fl]
                if (dx0 is None) != (dxr is None) or not np.array_equal(mask0, mask):
                    raise _Reject("the replayed pass touches other gradients than the recorded one")
                ok = ok & (span0 == span).all()
                if dx0 is not None:
                    ok = ok & (dx0 == dxr).all()
            # bit-for-bit verdict as ONE device scalar, read at the program's next sighting: reading it here would drain the host's whole lead
            # over the GPU (10+ ms of queued launches) for every recorded geometry
            prog.verdict = ok
    finally:
        ops.TAPE = tape
        ops.DEFER_REDUCE, ops.SIDE_WGRAD = flags
        ops.join_side_stream()        # (a rejected attempt may have queued sums into its scratch set: run them while the set is alive)
        restore_bn()
    return prog


def forward(net, fn, scope, x, eligible=True):
    """`fn(x)` - the forward of the sub-network whose parameters and buffers `net` (an nn.Module) holds - through a recorded program, or None
    (the caller runs the eager path). The sub-network must be a pure function of x's geometry: no random draws, no host decisions."""
    if not (ENABLED and eligible and _FUNCS and net.training and ops.PROF_SHAPES is None and x.is_cuda and x.dtype == torch.float32):
        return None
    if not (ops.TAPE is not None or torch.is_grad_enabled()):
        return None
    params = [p for p in net.parameters()]
    if not params or any(getattr(p, "_hwg_flat", None) is None for p in params):
        return None
    needs_dx = _needs_input_grad(x)
    key = (id(net), tuple(x.shape), bool(needs_dx), tuple(bool(p.requires_grad) for p in params), ops.TUNE_EPOCH, x.device.index,
           tuple(sorted(BACKWARD_FLAGS)))
    prog = _programs.get(key, 0)
    if prog is None:
        STATS["eager"] += 1
        return None
    if isinstance(prog, int):
        # (the eager sightings before a recording also warm the weight-image and plan caches)
        if prog + 1 < (RECORD_AFTER_DX if needs_dx else RECORD_AFTER) or sum(1 for v in _programs.values() if isinstance(v, Program)) >= MAX_PROGRAMS:
            _programs[key] = prog + 1
            STATS["eager"] += 1
            return None
        if ops._defer["count"] or ops._sn_defer or ops._side_dirty or ops.DEFER_KEEP_ARENA:
            return None                   # queued work of the surrounding pass: record at a quieter moment
        try:
            prog = _record(net, fn, scope, x, needs_dx, params)
            STATS["captures"] += 1
        except Exception as e:  # noqa: BLE001 - _Reject, or anything a recording trips over: the eager path is always there, in-process
            import warnings
            warnings.warn("handwriting_line_generation_amd.replay: %s pass %s stays on the eager path (%s: %s)" % (scope, tuple(x.shape), type(e).__name__, e))
            prog = None
            STATS["rejected"] += 1
        _programs[key] = prog
        STATS["eager"] += 1
        return None                       # this pass runs eagerly; the program's self-check is read at its next sighting
    if prog.verdict is not True:
        ok = bool(prog.verdict)
        prog.verdict = True
        if not ok:
            import warnings
            warnings.warn("handwriting_line_generation_amd.replay: %s pass %s stays on the eager path (the replayed pass does not reproduce the "
                          "recorded one bit for bit)" % (scope, tuple(x.shape)))
            prog.pools.clear()
            _programs[key] = None
            STATS["rejected"] += 1
            STATS["captures"] -= 1
            return None
    return _ReplayNet.apply(x, prog, *params)


def hwr_forward(net, x):
    """the recogniser (model/cnn_only_hwr.py): frozen weights, BatchNorm in train mode, no random draws"""
    return forward(net, net._forward, "HWR", x, eligible=net.logit_offset is None)
