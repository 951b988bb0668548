"""torch.autograd wrappers around the hwg C-ABI kernels.

All activations handled here are NHWC fp32 CUDA(HIP) tensors ([N,H,W,C], C fastest). Parameters keep the
PyTorch layouts of the reference's state-dict (Conv [O,I,R,S], ConvTranspose [I,O,R,S], Linear [O,I]) so
released checkpoints load; the kernels read/write those layouts directly through strides.

Nothing in this module computes on the CPU or through ATen math kernels: every op is a call into
libhwg_hip.so on torch's current stream. torch is used for memory, autograd bookkeeping and views.
"""
import ctypes
import math
import os

import torch

from . import _lib as L


class Function(torch.autograd.Function):
    """torch.autograd.Function with `.apply` bound straight to the C++ entry point. torch's Python-level `Function.apply` first walks the
    arguments for functorch wrappers and checks for a setup_context override - 10-25 us per call on the host, several times what the
    forward of a small op costs here, for ~140 ops per training step. None of these ops is used under functorch transforms."""

    # how `backward` treats S gradient sets stacked along the batch axis (Tape.backward_sets): "loop" = called once per set on that set's
    # slice (ops whose backward reads saved per-sample state), "stacked" = called once on the stacked gradient (stateless in the batch),
    # "native" = the class implements backward_sets(ctx, S, targets, *stacked_grads) itself
    sets = "loop"

    @classmethod
    def apply(cls, *args):
        if TAPE is not None:
            return TAPE.record(cls, args)
        return super(torch.autograd.Function, cls).apply(*args)


# ---- explicit tape (a sub-network run outside autograd so that SEVERAL upstream gradients can go through it in one pass) ----------------
# The balanced GAN lessons send two or three different gradients through the generator, one backward pass each, on layers that fill a
# quarter of the chip at 8 lines. With the generator's forward recorded on a Tape its backward can take the S gradients stacked along the
# batch axis: data-gradient convolutions, blurs and resamplings run ONCE on S x N samples; ops that read saved per-sample state (AdaIN,
# weight gradients, the style MLP) run per set on slices, each set accumulating its parameter gradients into its own buffer (grad_set).
TAPE = None


class _TapeCtx:
    """the part of torch.autograd's ctx the Functions of this module use"""

    def __init__(self, needs):
        self.needs_input_grad = needs
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def mark_non_differentiable(self, *a):
        pass

    def set_materialize_grads(self, v):
        pass


class SetGrad:
    """gradient of one taped tensor for S sets: `stacked` [S*N, ...] (sets along the batch axis) and / or `parts` (one tensor per set)"""

    __slots__ = ("stacked", "parts", "S")

    def __init__(self, S, stacked=None, parts=None):
        self.S, self.stacked, self.parts = S, stacked, parts

    def get_parts(self):
        if self.parts is None:
            n = self.stacked.shape[0] // self.S
            self.parts = [self.stacked[i * n:(i + 1) * n] for i in range(self.S)]
        return self.parts

    def get_stacked(self):
        if self.stacked is None:
            self.stacked = self.parts[0] if self.S == 1 else torch.cat([p.contiguous() for p in self.parts], dim=0)
        return self.stacked

    def add(self, other):
        a, b = self.get_parts(), other.get_parts()
        out = []
        for x, y in zip(a, b):
            z = torch.empty_like(x)
            L.call("hwg_axpby", x.contiguous(), 1.0, y.contiguous(), 1.0, z, x.numel(), _stream())
            out.append(z)
        return SetGrad(self.S, parts=out)


class Tape:
    def __init__(self):
        self.nodes = []          # (Function class | "alias", ctx, inputs (the original argument tuple), outputs tuple)
        self.live = {}           # id(tensor) -> tensor, for every tensor a gradient can arrive at (keeps the ids valid)
        self.by_mem = {}         # (address, elements) of the contiguous live tensors -> tensor (how reshapes of them are recognised)

    def watch(self, t):
        self._live(t)
        return t

    def _adopt_views(self, args):
        """torch-level reshapes of taped tensors (`y.view(rows, O)`, `.reshape(B, -1)`, also of results that are themselves views): autograd
        would track them; here a contiguous tensor that covers exactly the memory of a contiguous live tensor becomes an alias node. Any
        other view of a live tensor is an error."""
        for a in args:
            if not isinstance(a, torch.Tensor) or id(a) in self.live or a._base is None:
                continue
            base = self.by_mem.get((a.data_ptr(), a.numel())) if a.is_contiguous() else None
            if base is not None:
                self.live[id(a)] = a
                self.nodes.append(("alias", tuple(base.shape), (base,), (a,)))
            elif (a._base.data_ptr(), a._base.numel()) in self.by_mem:
                raise L.HwgError("taped forward: unsupported view of a taped tensor (shape %s of base %s)" % (tuple(a.shape), tuple(a._base.shape)))

    def _live(self, t):
        self.live[id(t)] = t
        if t.is_contiguous():
            self.by_mem.setdefault((t.data_ptr(), t.numel()), t)

    def record(self, cls, args):
        global TAPE
        self._adopt_views(args)
        needs = tuple(isinstance(a, torch.Tensor) and (id(a) in self.live or (a.requires_grad and a.is_leaf)) for a in args)
        ctx = _TapeCtx(needs)
        TAPE = None             # ops called from inside a forward are part of it, not nodes of their own
        try:
            out = cls.forward(ctx, *args)
        finally:
            TAPE = self
        if any(needs):
            outs = out if isinstance(out, tuple) else (out,)
            for o in outs:
                if isinstance(o, torch.Tensor):
                    self._live(o)
            self.nodes.append((cls, ctx, args, outs))
        return out

    def alias(self, new, old):
        """`new` is a reshape of the taped tensor `old` (same elements, same order)"""
        if id(old) in self.live:
            self._live(new)
            self.nodes.append(("alias", tuple(old.shape), (old,), (new,)))
        return new

    def backward_sets(self, out, grads, targets):
        """send S gradients of `out` (list of [N, ...] tensors) through the recorded ops in one pass. targets[s]: where set s accumulates its
        parameter gradients - None = the parameters' own .grad, else (flat buffer, touched mask) as produced by FlatParams.stash().
        -> {id(watched input): list of S gradient tensors}"""
        global GRAD_SET
        S = len(grads)
        g = {id(out): SetGrad(S, parts=[x.contiguous() for x in grads])}
        prev = GRAD_SET
        try:
            for cls, ctx, args, outs in reversed(self.nodes):
                gouts = [g.pop(id(o), None) if isinstance(o, torch.Tensor) else None for o in outs]
                if all(x is None for x in gouts):
                    continue
                if cls == "alias":      # ctx = the base tensor's shape; per set: the same elements in the base's shape
                    gin = [SetGrad(S, parts=[p_.contiguous().reshape(ctx) for p_ in gouts[0].get_parts()])]
                else:
                    for i, x in enumerate(gouts):      # a multi-output op with a missing output gradient: zeros, as autograd materialises them
                        if x is None and isinstance(outs[i], torch.Tensor):
                            z = torch.zeros((S * outs[i].shape[0],) + tuple(outs[i].shape[1:]), dtype=outs[i].dtype, device=outs[i].device)
                            gouts[i] = SetGrad(S, stacked=z)
                    gin = self._node_backward(cls, ctx, gouts, S, targets)
                for a, need, gi in zip(args, ctx.needs_input_grad if cls != "alias" else (True,), gin):
                    if gi is None or not need:
                        continue
                    if a.requires_grad and a.is_leaf and id(a) not in self.live:
                        # a parameter whose gradient came back as a tensor (not accumulated in place by the kernel): add it to its set's buffer
                        for s_, part in enumerate(gi.get_parts()):
                            GRAD_SET = targets[s_]
                            buf = _grad_buffer(a)
                            L.call("hwg_axpby", part.contiguous(), 1.0, buf, 1.0, buf, buf.numel(), _stream())
                        continue
                    k = id(a)
                    g[k] = gi if k not in g else g[k].add(gi)
        finally:
            GRAD_SET = prev
        return {k: v.get_parts() for k, v in g.items()}

    @staticmethod
    def _node_backward(cls, ctx, gouts, S, targets):
        global GRAD_SET
        mode = cls.sets if S > 1 else "loop"
        if mode == "native":
            res = cls.backward_sets(ctx, S, targets, *[x.get_stacked() for x in gouts])
            return [None if r is None else (r if isinstance(r, SetGrad) else SetGrad(S, stacked=r)) for r in res]
        if mode == "stacked":
            GRAD_SET = targets[0]
            res = cls.backward(ctx, *[x.get_stacked() for x in gouts])
            res = res if isinstance(res, tuple) else (res,)
            return [None if r is None else SetGrad(S, stacked=r) for r in res]
        per_set = []
        for s_ in range(S):
            GRAD_SET = targets[s_]
            res = cls.backward(ctx, *[x.get_parts()[s_] for x in gouts])
            per_set.append(res if isinstance(res, tuple) else (res,))
        out = []
        for i in range(len(per_set[0])):
            col = [r[i] for r in per_set]
            out.append(None if col[0] is None else SetGrad(S, parts=col))
        return out


# ---- debug guard for taped forwards (HWG_TAPE_CHECK=1) ---------------------------------------------------------------------------------
# A taped forward runs under no_grad and tracks only Function calls plus contiguous whole-memory reshapes of their results. Any OTHER
# torch-level op on a taped tensor (arithmetic, torch.cat, indexing, a norm in eval mode ...) would produce a tensor the tape does not
# know, and the gradient path through it would be dropped without an error. With the check on, every torch-level call made while a tape is
# recording is inspected: an argument that is (a view of) a taped tensor is only allowed in the shape / view / bookkeeping calls below.
TAPE_CHECK = bool(int(os.environ.get("HWG_TAPE_CHECK", "0") or 0))
_TAPE_SAFE = {"view", "reshape", "contiguous", "flatten", "size", "dim", "numel", "data_ptr", "is_contiguous", "stride", "detach", "requires_grad_",
              "__get__", "__len__", "__repr__", "storage_offset", "element_size", "untyped_storage", "is_floating_point", "get_device", "_is_view",
              "__set__", "__bool__", "is_leaf", "__format__", "__str__", "type", "nelement", "ndimension", "__hash__", "__reduce_ex__"}


class _TapeGuard(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        tape = TAPE
        if tape is not None:                       # (None inside an op's own forward: ops allocate and launch freely there)
            name = getattr(func, "__name__", str(func))
            if name not in _TAPE_SAFE:
                stack = list(args) + list((kwargs or {}).values())
                while stack:
                    a = stack.pop()
                    if isinstance(a, (list, tuple)):
                        stack.extend(a)
                    elif isinstance(a, torch.Tensor) and (id(a) in tape.live or (a._base is not None and id(a._base) in tape.live)):
                        raise L.HwgError("taped forward: torch-level op %r on a taped tensor (shape %s) - its result would be unknown to the tape and the "
                                         "gradient through it silently dropped; route it through an ops.Function" % (name, tuple(a.shape)))
        return func(*args, **(kwargs or {}))


class taping:
    """`with ops.taping(tape):` - Function.apply records on `tape` instead of autograd's graph (no_grad is the caller's business)"""

    def __init__(self, tape):
        self.tape, self.guard = tape, (_TapeGuard() if TAPE_CHECK else None)

    def __enter__(self):
        global TAPE
        TAPE = self.tape
        if self.guard is not None:
            self.guard.__enter__()
        return self.tape

    def __exit__(self, *exc):
        global TAPE
        if self.guard is not None:
            self.guard.__exit__(*exc)
        TAPE = None
        return False


ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3
NORM_IN, NORM_GN, NORM_BN = 0, 1, 2


_raw_stream = torch._C._cuda_getCurrentRawStream if hasattr(torch._C, "_cuda_getCurrentRawStream") else None
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    # raw hipStream_t of torch's current stream (the Python Stream object is ~20 us to build, torch.cuda.current_device() re-checks the lazy
    # initialisation on every call: ~1 us; the two C entry points together ~0.3 us - this runs ~430 times per training step)
    if _raw_stream is None or _get_device is None:       # a torch build without the private entry points: the public (slower) API
        return torch.cuda.current_stream().cuda_stream
    return _raw_stream(_get_device())


_copy_streams = {}


class AsyncFetch:
    """Device->host copy on a side stream: the producer's stream keeps running (and the host keeps enqueueing) while the bytes
    travel; `.get()` waits only for the copy. Replaces `.cpu()` / `.item()` where the value is needed later than it is produced."""

    def __init__(self, tensor):
        dev = tensor.device
        cs = _copy_streams.get(dev)
        if cs is None:
            cs = _copy_streams[dev] = torch.cuda.Stream(device=dev)
        self.src = tensor.detach()
        self.host = torch.empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(cs):
            cs.wait_event(ready)
            self.host.copy_(self.src, non_blocking=True)
            self.done = torch.cuda.Event()
            self.done.record()
        self.src.record_stream(cs)

    def get(self):
        self.done.synchronize()
        return self.host


def h2d(host, device, dtype=None):
    """Asynchronous host->device upload through pinned staging memory. A plain `.to(device)` of pageable memory blocks the host
    until every kernel queued before it has finished, which serialises the CPU against the GPU a dozen times per step."""
    t = torch.as_tensor(host)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.is_cuda:
        return t.to(device)
    pinned = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    pinned.copy_(t)
    return pinned.to(device, non_blocking=True)


_ws_cache = {}


def workspace(nbytes, device):
    """Shared scratch buffer (stream ordered, contents are dead once the call that used it returns)."""
    nbytes = int(nbytes)
    # one buffer per (device, stream): passes that run concurrently on different streams (the per-set style passes of the GAN trainer) must
    # not share scratch memory; torch's allocator ties the buffer to the stream that is current when it is created
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# ---- side stream for weight gradients -----------------------------------------------------------------------------------
# Nothing in the backward pass waits for a conv's weight gradient, only the optimizer side does. With SIDE_WGRAD on (the trainers
# switch it on and call join_side_stream() after every backward()), the MFMA weight-gradient kernels are enqueued on a second HIP
# stream: their workgroups fill the CUs that the data-gradient chain leaves idle (kernel tails, small elementwise launches).
# Per tensor the accumulation order is unchanged (all weight gradients run in order on the one side stream).
SIDE_WGRAD = False
_side_streams = {}
_side_dirty = set()


def _side_stream(device):
    key = device.index if device.index is not None else torch.cuda.current_device()
    hit = _side_streams.get(key)
    if hit is None:
        st = torch.cuda.Stream(device=device)
        hit = _side_streams[key] = (st, st.cuda_stream, key)
    return hit


def _side_workspace(nbytes, device, stream):
    key = ("side", device.index if device.index is not None else torch.cuda.current_device())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        buf.record_stream(stream)
        _ws_cache[key] = buf
    return buf


_side_hold = []      # operands of side-stream launches: kept alive until the join so that the allocator cannot hand their memory to a main-stream kernel


def join_side_stream():
    """make torch's current stream wait for every weight gradient enqueued on the side stream so far, then sum the weight-gradient partial
    images whose reduction was deferred (flush_deferred_reduce): after this call every parameter gradient is complete on the current stream"""
    for key in list(_side_dirty):
        L.call("hwg_stream_join", _side_streams[key][1], _stream())
    _side_dirty.clear()
    del _side_hold[:]
    if _defer["count"] or _sn_defer:
        flush_deferred_reduce()


# ---- deferred sum of the weight-gradient partial images -------------------------------------------------------------------
# A weight gradient leaves one partial image per pixel range in its workspace and sums them with a launch of its own: 65 launches of
# 5..25 us per training step, too small to stream at memory rate. With DEFER_REDUCE on (the trainers switch it on around their backward
# passes and call join_side_stream() after every backward()), a weight gradient's workspace comes out of a private arena instead of the
# shared scratch buffer, its sum is only queued (hwg_wgrad_defer_next), and flush_deferred_reduce() sums everything queued with one
# table-driven launch per 32 gradients. Bit-identical to the undeferred order (same schedule per gradient, several gradients of one tensor in
# queue order). Gradients are undefined between the backward call and the flush.
DEFER_REDUCE = False
_defer = {"count": 0, "offset": 0, "arena": {}, "launches": 0, "flushes": 0, "fallbacks": 0, "sn_launches": 0}
_sn_defer = []       # queued spectral-norm backward passes: (dW_sn, W_bar, u, v, sigma, destination, R, K), tensors held until the flush
_SN_BWD_REC = [("dWsn", "<u8"), ("Wbar", "<u8"), ("u", "<u8"), ("v", "<u8"), ("sigma", "<u8"), ("dst", "<u8"), ("R", "<i4"), ("K", "<i4"), ("acc", "<i4"), ("pad", "<i4")]
DEFER_ARENA_BYTES = int(os.environ.get("HWG_DEFER_ARENA_MB", "1536")) << 20


def _defer_workspace(nbytes, device):
    """`nbytes` of the arena that stays untouched until the next flush, or None when the arena is full (the caller then sums at once)"""
    key = device.index if device.index is not None else torch.cuda.current_device()
    arena = _defer["arena"].get(key)
    if arena is None:
        arena = _defer["arena"][key] = torch.empty(DEFER_ARENA_BYTES, dtype=torch.uint8, device=device)
        for st in _side_streams.values():
            arena.record_stream(st[0])
    off = _defer["offset"]
    need = (int(nbytes) + 255) & ~255
    if off + need > arena.numel():
        _defer["fallbacks"] += 1
        return None
    _defer["offset"] = off + need
    _defer["count"] += 1
    return arena[off: off + need]


DEFER_KEEP_ARENA = False     # passes on several streams in flight: a flush must not hand the arena's start to the next pass (reset after the join)


_FLUSH_LAUNCHES = None     # the flush's `int* launches` out-parameter: a module-lifetime host array (never a temporary: a call list that
                            # recorded the address of a temporary would write through a dangling pointer on every replay)


def flush_deferred_reduce():
    """sum every queued set of partial images on the current stream (call after the side stream has been joined)"""
    global _FLUSH_LAUNCHES
    import numpy as np
    if _FLUSH_LAUNCHES is None:
        _FLUSH_LAUNCHES = np.zeros(1, dtype=np.int32)
    n = _FLUSH_LAUNCHES
    n[0] = 0
    L.call("hwg_wgrad_defer_flush", _stream(), n.ctypes.data)
    _defer["launches"] += int(n[0]); _defer["flushes"] += 1
    _defer["count"] = 0
    if not DEFER_KEEP_ARENA:
        _defer["offset"] = 0
    if _sn_defer:          # queued spectral-norm backward passes (their dW_sn inputs are complete: summed eagerly, or by the launch above)
        # a launch adds every entry's result to its destination with a plain read-modify-write: two entries with the SAME destination (the
        # discriminator applied twice in one backward pass - a lesson with both 'disc' and 'gen') must not share a launch. The k-th
        # occurrence of a destination goes to pass k, passes run one after the other in queue order (as hwg_wgrad_defer_flush does for convs).
        passes, seen = [], {}
        for e in _sn_defer:
            k = seen.get(e[5].data_ptr(), 0)
            seen[e[5].data_ptr()] = k + 1
            if k == len(passes):
                passes.append([])
            passes[k].append(e)
        chunks = [pas[i: i + 16] for pas in passes for i in range(0, len(pas), 16)]
        for chunk in chunks:
            rec = np.zeros(len(chunk), dtype=_SN_BWD_REC)
            for k, (dwsn, w_bar, u, v, sigma, dst, R, K) in enumerate(chunk):
                rec[k] = (dwsn.data_ptr(), w_bar.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(), dst.data_ptr(), R, K, 1, 0)
            ws = workspace(L.query("hwg_spectral_bwd_multi_workspace", len(chunk)), chunk[0][0].device)
            L.call("hwg_spectral_bwd_multi", rec.ctypes.data, len(chunk), ws, ws.numel(), _stream())
            _defer["sn_launches"] += 1
        del _sn_defer[:]


def reset_defer_arena():
    _defer["offset"] = 0


def _chk(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise L.HwgError("%s must live on the GPU (no CPU fallback exists)" % name)
    if t.dtype != dtype:
        raise L.HwgError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise L.HwgError("%s must be contiguous" % name)


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


# Parameter gradients are accumulated by the kernels straight into `param.grad` (the trainer keeps those as views of one flat
# buffer) instead of being returned to autograd, which would allocate a temporary and launch an extra add kernel per tensor.
DIRECT_PARAM_GRADS = True


def _direct(p):
    return DIRECT_PARAM_GRADS and p is not None and p.is_leaf and p.requires_grad


# Gradient-set redirect: None = parameter gradients accumulate into param.grad (views of the trainer's flat buffer). Otherwise
# (buffer, touched mask) of a stashed gradient set (FlatParams.stash()): kernels accumulate into the SAME offsets of that buffer and the
# set's mask is marked - how the batched generator backward (Tape.backward_sets) and the per-set style-extractor passes put every set's
# gradients where the reference's sequential backward + clone-and-zero would have put them.
GRAD_SET = None
_set_views = {}


class grad_set:
    def __init__(self, target):
        self.target = target

    def __enter__(self):
        global GRAD_SET
        self.prev = GRAD_SET
        GRAD_SET = self.target

    def __exit__(self, *exc):
        global GRAD_SET
        GRAD_SET = self.prev


def _grad_buffer(p):
    """the buffer parameter gradients accumulate into (zero-initialised on first use)"""
    gs = GRAD_SET
    if gs is not None:
        info = getattr(p, "_hwg_flat", None)
        if info is None:
            raise L.HwgError("gradient-set redirect needs parameters that live in a FlatParams buffer")
        flat, k = info
        buf, mask = gs
        mask[k] = True
        key = (buf.data_ptr(), id(flat), k)
        v = _set_views.get(key)
        if v is None:
            off = int(flat.offsets[k])
            v = _set_views[key] = buf[off: off + int(flat.numel[k])].view_as(p)
        return v
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    touch = getattr(p, "_hwg_touch", None)
    if touch is not None:
        touch()
    return p.grad


# ----------------------------------------------------------------------------------------------
# convolution
# ----------------------------------------------------------------------------------------------
def _desc(N, H, W, C, K, R, S, stride, pad, dil, P, Q, transposed=0):
    d = L.ConvDesc()
    d.N, d.H, d.W, d.C, d.K, d.R, d.S = N, H, W, C, K, R, S
    d.stride_h, d.stride_w = stride
    d.pad_h, d.pad_w = pad
    d.dil_h, d.dil_w = dil
    d.P, d.Q = P, Q
    d.transposed = transposed
    return d


# Packed ([tap][K][C]) images of leaf parameters are cached until the parameter changes: `_version` catches torch-side writes
# (load_state_dict, copy_), WEIGHT_EPOCH[...] is bumped by the multi-tensor Adam kernel, which writes through raw pointers.
# After an optimizer step `repack_group` refreshes every cached image of that optimizer's parameters in ONE launch
# (hwg_conv_pack_weight_multi) instead of ~100 single-weight launches spread over the next step.
WEIGHT_EPOCH = {}
_pack_cache = {}          # key -> [version, epoch, packed tensor, weight, (A, B, Bpad, R, S, sa, sb, flip)]
_pack_tables = {}         # group -> (number of cache entries it was built for, device table, total blocks, entries)
PACK_PER_BLOCK = 1024
_PACK_DTYPE = None


def bump_weight_epoch(group):
    WEIGHT_EPOCH[group] = WEIGHT_EPOCH.get(group, 0) + 1


def _pack(weight, A, B, R, S, sa, sb, flip, Bpad=None, wino=False):
    """engine image of a weight: [tap][A][Bpad], or (wino) the Winograd-domain filters U = G g G^T as [Bpad/16][16][Apad][16]"""
    Bpad = B if Bpad is None else Bpad
    Apad = (A + 15) // 16 * 16 if wino else A
    if wino == 2:
        # F(3x3,2x2) image of a 4x4 stride-2 convolution's weight (hwg_wino_s2_pack_weight): A = the convolution's output channels, B = its input
        # channels, sa / sb their strides, flip = 1: the image of the data-gradient product ((a,b,c) block channels x K)
        Apad, Bpad = ((4 * B + 15) // 16 * 16, (A + 15) // 16 * 16) if flip else ((A + 15) // 16 * 16, (4 * B + 15) // 16 * 16)
    elif wino:
        Bpad = (B + 15) // 16 * 16
    pre = getattr(weight, "_hwg_prepack", None)
    if pre is not None:
        # a spectral-norm layer's W_bar / sigma: its images were written by the network's one scaled multi-pack launch (SpectralBank.update);
        # an image nobody asked for before is packed here, from the tensor itself, and joins the launch from the next forward pass on
        variant = (A, B, Bpad, R, S, sa, sb, int(flip), int(wino), Apad)
        img = pre.images.get(variant)
        if img is not None:
            return img
        pre.miss(variant, weight)
    key = hit = group = None
    # cached per PARAMETER only: under no_grad (taped forwards, the discriminator lessons' detached generator pass) every derived weight - the
    # fused-upsample 4x4 image, an equal-lr scaled copy - is a "leaf" too, and caching those kept one entry (the temporary and its packed image)
    # alive per forward pass for the rest of the run
    if isinstance(weight, torch.nn.Parameter):
        group = getattr(weight, "_hwg_group", None)
        epoch = WEIGHT_EPOCH.get(group, 0)
        key = (id(weight), weight.data_ptr(), A, B, Bpad, sa, sb, int(flip), int(wino))
        hit = _pack_cache.get(key)
        if hit is not None and hit[0] == weight._version and hit[1] == epoch:
            return hit[2]
    if key is not None and hit is not None:
        out = hit[2]
    elif wino:
        out = torch.empty((Bpad // 16, 16, Apad, 16), dtype=torch.float32, device=weight.device)
    else:
        out = torch.empty((R * S, A, Bpad), dtype=torch.float32, device=weight.device)
    if wino == 2:
        L.call("hwg_wino_s2_pack_weight", weight, out, A, B, sa, sb, int(flip), _stream())
    elif wino:
        L.call("hwg_wino_pack_weight", weight, out, A, B, sa, sb, S, 1, int(flip), _stream())
    else:
        L.call("hwg_conv_pack_weight", weight, out, A, B, Bpad, R, S, sa, sb, S, 1, int(flip), _stream())
    if key is not None:
        _pack_cache[key] = [weight._version, epoch, out, weight, (A, B, Bpad, R, S, sa, sb, int(flip), int(wino), Apad), group]
    return out


def repack_group(group):
    """refresh all cached packed images of the parameters of `group` (call right after bump_weight_epoch(group))"""
    global _PACK_DTYPE
    import numpy as np
    entries = [e for e in _pack_cache.values() if e[5] == group]
    if not entries:
        return
    tab = _pack_tables.get(group)
    if tab is None or tab[0] != len(entries):
        if _PACK_DTYPE is None:
            _PACK_DTYPE = np.dtype([("src", "<u8"), ("dst", "<u8"), ("A", "<i4"), ("B", "<i4"), ("Bpad", "<i4"), ("R", "<i4"), ("S", "<i4"), ("flip", "<i4"),
                                    ("sa", "<i8"), ("sb", "<i8"), ("sr", "<i8"), ("ss", "<i8"), ("total", "<i8"), ("first_block", "<i8"),
                                    ("mode", "<i4"), ("Apad", "<i4")])
        host = np.zeros(len(entries), dtype=_PACK_DTYPE)
        blocks = 0
        for i, e in enumerate(entries):
            A, B, Bpad, R, S, sa, sb, flip, mode, Apad = e[4]
            total = Apad * Bpad if mode else R * S * A * Bpad
            host[i] = (e[3].data_ptr(), e[2].data_ptr(), A, B, Bpad, R, S, flip, sa, sb, S, 1, total, blocks, mode, Apad)
            blocks += (total + PACK_PER_BLOCK - 1) // PACK_PER_BLOCK
        dev = h2d(torch.from_numpy(host.view(np.uint8)), entries[0][2].device)
        tab = _pack_tables[group] = (len(entries), dev, blocks, entries)
    L.call("hwg_conv_pack_weight_multi", tab[1], tab[0], tab[2], _stream())
    epoch = WEIGHT_EPOCH.get(group, 0)
    for e in tab[3]:
        if e[0] == e[3]._version:    # torch-side writes still force a private re-pack on next use
            e[1] = epoch


def _pad_channels(x, Cpad):
    N, H, W, C = x.shape
    out = torch.empty((N, H, W, Cpad), dtype=x.dtype, device=x.device)
    L.call("hwg_pad_channels", x, C, out, Cpad, N * H * W, _stream())      # copy + zero lanes in one launch (and visible to replay.py's call log)
    return out


# Per-launch timing of the MFMA conv kernels (bench.py's roofline measurement) is done inside the library (hwg_prof_*: HIP events
# recorded right around the launches). Here we only label the launches with their layer shape when a profile is running.
PROF_SHAPES = None     # None, or {shape tuple: tag}
SCOPE = None           # which network is running its forward ("G", "D", ...): convs remember it so that a profile can be split by network


class scope:
    """`with ops.scope("G"):` around a network's forward; nested scopes keep the outermost label"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global SCOPE
        self.prev = SCOPE
        if SCOPE is None:
            SCOPE = self.name

    def __exit__(self, *exc):
        global SCOPE
        SCOPE = self.prev


def prof_start(max_records=200000):
    global PROF_SHAPES
    PROF_SHAPES = {}
    L.call("hwg_prof_start", max_records)


def prof_enable(on):
    """pause / resume an open profile"""
    L.call("hwg_prof_enable", int(bool(on)))


def prof_stop():
    """-> list of (kind, shape, work, seconds); kind in conv_mfma_kernel / wgrad_mfma_kernel / conv_split_reduce / wgrad_reduce / direct kernels /
    wino_conv_kernel (the three Winograd forward / data-gradient kernels; work = direct-form FLOPs of the layer) / wino_wgrad_kernel"""
    global PROF_SHAPES
    import numpy as np
    cap = 200000
    kinds = np.zeros(cap, dtype=np.int32); tags = np.zeros(cap, dtype=np.int32)
    work = np.zeros(cap, dtype=np.float64); ms = np.zeros(cap, dtype=np.float32)
    n = L.query("hwg_prof_stop", kinds.ctypes.data, tags.ctypes.data, work.ctypes.data, ms.ctypes.data, cap)
    by_tag = {v: k for k, v in (PROF_SHAPES or {}).items()}
    PROF_SHAPES = None
    names = ("conv_mfma_kernel", "wgrad_mfma_kernel", "conv_split_reduce_kernel", "wgrad_reduce_kernel", "conv_direct_kernels", "wgrad_direct_kernels",
             "wino_conv_kernel", "wino_wgrad_kernel")
    return [(names[kinds[i]], by_tag.get(int(tags[i])), float(work[i]), float(ms[i]) * 1e-3) for i in range(n)]


def _prof_tag(shape):
    tag = PROF_SHAPES.get(shape)
    if tag is None:
        tag = PROF_SHAPES[shape] = len(PROF_SHAPES)
    L.call("hwg_prof_tag", tag)


_RUN_SCOPE = [None]     # scope of the conv op currently being executed (forward: SCOPE, backward: the scope saved at forward time)


_conv_plans = {}    # geometry -> (descriptor, workspace bytes): descriptors are built (and the library's schedule queried) once per geometry


def static_host_ptrs():
    """addresses of the host-side descriptors that live as long as the plan caches (what a recorded call list may hold as literal host
    pointers - replay.py rejects every other host address: a temporary's address would dangle on replay)"""
    out = {int(p[2]) for p in _conv_plans.values()}
    out.update(int(p[0].ptr) for p in _wgrad_plans.values())
    return out


def _run_conv(x, wp, bias, N, H, W, C, K, R, S, stride, pad, dil, P, Q, transposed, wino=False):
    y = torch.empty((N, P, Q, K), dtype=torch.float32, device=x.device)
    key = (N, H, W, C, K, R, S, stride, pad, dil, P, Q, transposed, wino)
    plan = _conv_plans.get(key)
    if plan is None:
        d = _desc(N, H, W, C, K, R, S, stride, pad, dil, P, Q, transposed)
        plan = _conv_plans[key] = (d, L.query(("hwg_conv_fwd_workspace", "hwg_wino_conv_workspace", "hwg_wino_s2_workspace")[int(wino)], d.ptr), d.ptr)
    d, need, dptr = plan
    if PROF_SHAPES is not None:
        _prof_tag((N, H, W, C, K, R, S, stride, pad, dil, transposed, _RUN_SCOPE[0]))
    ws = workspace(need, x.device) if need else None
    L.call(("hwg_conv_fwd", "hwg_wino_conv_fwd", "hwg_wino_s2_conv")[int(wino)], dptr, x, wp, bias, y, 0, ws, need, _stream())
    return y


# 3x3 / stride 1 / dilation 1 products with >= 16 output channels run as F(2x2,3x3) (csrc/conv_wino.hip); HWG_WINO=0 keeps them on the
# direct implicit-GEMM kernels (A/B timing and numerics comparisons)
WINOGRAD = bool(int(os.environ.get("HWG_WINO", "1") or 1))


_wino_choice = {}


TUNE_EPOCH = 0      # bumped by tuning_reload(): anything derived from the library's plans (recorded launch lists, replay.py) is keyed by it


def tuning_reload():
    """The library plans every geometry once and reads its tuning knobs (HWG_WINO, HWG_WINO_FORCE, HWG_CONV_FORCE, ...) at plan time; the
    per-geometry choices are cached here as well. Call this after changing such a variable in a running process."""
    global WINOGRAD, TUNE_EPOCH
    TUNE_EPOCH += 1
    WINOGRAD = bool(int(os.environ.get("HWG_WINO", "1") or 1))
    _wino_choice.clear(); _wino_wgrad_choice.clear(); _conv_plans.clear(); _wgrad_plans.clear(); _wgrad_sets_ok.clear(); _wgrad_sets_ws.clear()
    L.call("hwg_tuning_reload")


class tuning:
    """`with ops.tuning(HWG_WINO="2", HWG_WINO_FORCE="6"):` - set tuning variables for a block (tests, sweeps) and restore them after"""

    def __init__(self, **env):
        self.env = {k: (None if v is None else str(v)) for k, v in env.items()}

    def __enter__(self):
        self.prev = {k: os.environ.get(k) for k in self.env}
        for k, v in self.env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        tuning_reload()
        return self

    def __exit__(self, *exc):
        for k, v in self.prev.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        tuning_reload()


def last_plan():
    """(engine, schedule id, split factor) of this thread's last convolution-family launch (engine: the profiler's kind codes - 0 direct
    MFMA conv, 1 MFMA weight gradient, 6 Winograd conv, 7 Winograd weight gradient)"""
    import numpy as np
    out = np.full(3, -1, dtype=np.int32)
    L.call("hwg_last_plan", out.ctypes.data)
    return tuple(int(v) for v in out)


def _wino_ok(N, H, W, C, K, R, S, stride, pad, dil, P, Q):
    """the library's cost models decide per product (cached per geometry): many tiles x few channels run as Winograd, few tiles x many
    channels (the 512-channel recogniser layers) stay on the direct kernels"""
    if not (WINOGRAD and R == 3 and S == 3 and stride == (1, 1) and dil == (1, 1) and C % 16 == 0 and K >= 16 and pad[0] >= 0 and pad[1] >= 0):
        return False
    key = (N, H, W, C, K, pad)
    hit = _wino_choice.get(key)
    if hit is None:
        d = _desc(N, H, W, C, K, 3, 3, (1, 1), pad, (1, 1), P, Q, 0)
        hit = _wino_choice[key] = bool(L.query("hwg_wino_preferred", d.ptr))
    return hit


def _wino_s2_ok(N, H, W, C, K, R, S, stride, pad, dil, P, Q, transposed):
    """4x4 stride-2 pad-0 layers (transposed = 1: their data gradient, described as the fractionally strided product) in the Winograd domain
    F(3x3,2x2) on the space-to-depth image (csrc/conv_wino.hip, hwg_wino_s2_*) - the library's cost models decide per geometry"""
    if not (WINOGRAD and R == 4 and S == 4 and stride == (2, 2) and pad == (0, 0) and dil == (1, 1) and C % 16 == 0):
        return False
    key = (N, H, W, C, K, "s2", transposed, P, Q)     # (P, Q: the data gradient of an odd-width input is not the F(3x3,2x2) geometry)
    hit = _wino_choice.get(key)
    if hit is None:
        d = _desc(N, H, W, C, K, 4, 4, (2, 2), (0, 0), (1, 1), P, Q, transposed)
        hit = _wino_choice[key] = bool(L.query("hwg_wino_s2_preferred", d.ptr))
    return hit


_wino_wgrad_choice = {}


def _wino_wgrad_ok(d):
    """weight gradient of a 3x3 stride-1 layer in the Winograd domain (conv_wino_wgrad.hip) - the library decides per geometry"""
    if not WINOGRAD or not ((d.R == 3 and d.S == 3) or (d.R == 4 and d.S == 4 and d.stride_h == 2 and d.stride_w == 2 and not d.transposed)):
        return False      # (4x4 stride 2 pad 0: the two-tap problem on the space-to-depth image, F(2x2 taps, 3x3 gradient tiles))
    key = (d.N, d.H, d.W, d.C, d.K, d.R, d.stride_h, d.stride_w, d.pad_h, d.pad_w, d.dil_h, d.dil_w, d.P, d.Q)
    hit = _wino_wgrad_choice.get(key)
    if hit is None:
        hit = _wino_wgrad_choice[key] = bool(L.query("hwg_wino_wgrad_preferred", d.ptr))
    return hit


def _cpad(C, K, fractional=False):
    """channel count the contraction side must be zero-padded to: multiples of 16 for the MFMA path (always taken by the fractionally
    strided mode: conv-transpose with stride > 1, data gradient of a strided conv), multiples of 4 for the K <= 2 direct kernel,
    nothing for the single-channel direct kernel"""
    if fractional:
        return (C + 15) // 16 * 16
    if C == 1:
        return C
    if K <= 2:
        return (C + 3) // 4 * 4
    return (C + 15) // 16 * 16


def _taps(weight):
    d = weight.dim()
    if d == 4:
        return weight.shape[2], weight.shape[3]
    if d == 3:
        return 1, weight.shape[2]
    if d == 2:
        return 1, 1
    raise L.HwgError("conv weight must have 2, 3 or 4 dimensions")


_wgrad_plans = {}
_wgrad_sets_ok = {}
_wgrad_sets_ws = {}


def _make_wgrad_plan(N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, P, Q, transposed):
    """(descriptor, engine, workspace bytes, Kq, Cq, tiny_end, tap_gemm) of a weight gradient; engine 0 = Winograd F(3x3,2x2),
    1 = MFMA kernel on channel-padded copies, 2 = MFMA / direct kernel on the tensors as they are"""
    if not transposed:
        d = _desc(N, H, W, C, K, R, S, (sh, sw), (ph, pw), (dh, dw), P, Q)
    else:
        d = _desc(N, P, Q, K, C, R, S, (sh, sw), (ph, pw), (dh, dw), H, W)
    Kq, Cq = (d.K + 3) // 4 * 4, (d.C + 3) // 4 * 4
    # single-channel ends (C == 1 first layers, K <= 2 heads) have a direct kernel; on large images the MFMA kernel on a
    # 4-channel zero-padded copy is faster (measured 164 vs 390 us for the 7x7 first layer of D), so only small ones stay direct
    # (the direct kernel runs a K<=2 head as ONE workgroup: 166 us for the discriminator's 256->1 3x3 head at 304 pixels, so heads
    # with a wide gathered side go through the padded MFMA path as well; only narrow-and-small cases stay direct)
    tiny_end = ((d.C == 1 and d.K % 4 == 0 and d.K > 2) or (d.K <= 2 and d.C % 4 == 0 and d.C < 16)) and d.N * d.P * d.Q < 8192
    # single gathered channel (first layers): the library runs the taps as the GEMM's N dimension, no padding needed
    tap_gemm = d.C == 1 and d.K > 2 and d.K % 4 == 0 and R * S <= 64
    if _wino_wgrad_ok(d):
        return d, 0, L.query("hwg_wino_wgrad_workspace", d.ptr), Kq, Cq, tiny_end, tap_gemm
    if (Kq != d.K or Cq != d.C) and not tiny_end and not tap_gemm:
        d.K, d.C = Kq, Cq
        return d, 1, L.query("hwg_conv_wgrad_workspace", d.ptr), Kq, Cq, tiny_end, tap_gemm
    return d, 2, L.query("hwg_conv_wgrad_workspace", d.ptr), Kq, Cq, tiny_end, tap_gemm


ONEROW_TWIN_DEAD_ROWS = bool(int(os.environ.get("HWG_ONEROW_DEAD_ROWS", "1") or 0))     # 0: inputs with dead rows (H > R, one output row) keep the sh-strided classes (A/B)


class _Conv2d(Function):
    """y = conv2d(x, w) (+b) or conv_transpose2d(x, w) (+b); x NHWC."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, transposed, output_padding):
        _chk(x, "conv input"); _chk(weight, "conv weight"); _chk(bias, "conv bias")
        ctx.scope = _RUN_SCOPE[0] = SCOPE
        N, H, W, C = x.shape
        # Linear [O,I] and Conv1d [O,I,S] parameters are used as they are (same memory as [O,I,1,S]); a .view() of the parameter would be a
        # non-leaf tensor, lose the packed-weight cache and the direct gradient accumulation and cost three extra ATen ops per backward
        R, S = _taps(weight)
        sh, sw = stride; ph, pw = padding; dh, dw = dilation
        if not transposed:
            K = weight.shape[0]
            assert weight.shape[1] == C, "conv: weight expects %d input channels, got %d" % (weight.shape[1], C)
            P = (H + 2 * ph - dh * (R - 1) - 1) // sh + 1
            Q = (W + 2 * pw - dw * (S - 1) - 1) // sw + 1
            Cp = _cpad(C, K)
            xin = _pad_channels(x, Cp) if Cp != C else x
            if Cp == C and _wino_s2_ok(N, H, W, C, K, R, S, (sh, sw), (ph, pw), (dh, dw), P, Q, 0):
                wp = _pack(weight, K, C, 4, 4, C * 16, 16, flip=0, wino=2)
                y = _run_conv(x, wp, bias, N, H, W, C, K, 4, 4, (2, 2), (0, 0), (1, 1), P, Q, 0, 2)
            else:
                wino = _wino_ok(N, H, W, Cp, K, R, S, (sh, sw), (ph, pw), (dh, dw), P, Q)
                wp = _pack(weight, K, C, R, S, C * R * S, R * S, flip=0, Bpad=Cp, wino=wino)
                y = _run_conv(xin, wp, bias, N, H, W, Cp, K, R, S, (sh, sw), (ph, pw), (dh, dw), P, Q, 0, wino)
        else:
            K = weight.shape[1]
            assert weight.shape[0] == C, "conv_transpose: weight expects %d input channels, got %d" % (weight.shape[0], C)
            oph, opw = output_padding
            if H == 1 and sh == 1 and ph == 0 and dh == 1 and dw == 1 and R > 1 and oph == 0:
                # ONE input row (the generator's 1-D -> 2-D lift, ConvTranspose (4,3)): output row p takes tap row r = p and nothing else, so
                # the layer is its own stride-(R, sw) twin. Run as that fractionally strided layer every output row is a parity class with
                # its 1 x S taps; as a stride-1 layer it is a correlation over R x S taps of which R - 1 rows only ever meet padding
                # (4 x the multiplies: 62 -> 25 us at 8 lines, 325 -> 100 us at a generation batch of 64). Backward follows ctx.geom.
                sh = R
                stride = (R, sw)
            P = (H - 1) * sh - 2 * ph + dh * (R - 1) + 1 + oph
            Q = (W - 1) * sw - 2 * pw + dw * (S - 1) + 1 + opw
            Cp = _cpad(C, K, fractional=(sh != 1 or sw != 1))
            xin = _pad_channels(x, Cp) if Cp != C else x
            if sh == 1 and sw == 1:
                # stride-1 transposed conv == correlation with mirrored taps and padding dil*(R-1)-pad
                wino = _wino_ok(N, H, W, Cp, K, R, S, (1, 1), (dh * (R - 1) - ph, dw * (S - 1) - pw), (dh, dw), P, Q)
                wp = _pack(weight, K, C, R, S, R * S, K * R * S, flip=1, Bpad=Cp, wino=wino)
                y = _run_conv(xin, wp, bias, N, H, W, Cp, K, R, S, (1, 1), (dh * (R - 1) - ph, dw * (S - 1) - pw), (dh, dw), P, Q, 0, wino)
            else:
                assert dh == 1 and dw == 1, "strided conv_transpose needs dilation 1"
                wp = _pack(weight, K, C, R, S, R * S, K * R * S, flip=0, Bpad=Cp)
                y = _run_conv(xin, wp, bias, N, H, W, Cp, K, R, S, (sh, sw), (ph, pw), (1, 1), P, Q, 1)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        ctx.param_refs = (weight, bias)
        ctx.geom = (stride, padding, dilation, transposed, P, Q)
        return y

    sets = "native"

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = _Conv2d._dgrad(ctx, dy) if ctx.needs_input_grad[0] else None
        dw_, db = _Conv2d._wgrad(ctx, dy)
        return dx, dw_, db, None, None, None, None, None

    @staticmethod
    def backward_sets(ctx, S, targets, dy):
        """S gradient sets stacked along the batch axis (Tape.backward_sets): ONE data-gradient launch over S x N samples (the generator's
        layers fill a quarter of the chip at 8 lines), one weight gradient per set (x is shared, every set accumulates into its own buffer)"""
        global GRAD_SET
        dy = dy.contiguous()
        dx = _Conv2d._dgrad(ctx, dy) if ctx.needs_input_grad[0] else None
        grouped = _Conv2d._wgrad_sets(ctx, dy, S, targets) if S > 1 else None
        if grouped is not None:
            dws, dbs = grouped
        else:
            n = dy.shape[0] // S
            dws, dbs = [], []
            for s_ in range(S):
                GRAD_SET = targets[s_]
                dw_, db = _Conv2d._wgrad(ctx, dy[s_ * n:(s_ + 1) * n])
                dws.append(dw_); dbs.append(db)
        return (dx, None if dws[0] is None else SetGrad(S, parts=dws), None if dbs[0] is None else SetGrad(S, parts=dbs), None, None, None, None, None)

    @staticmethod
    def _wgrad_sets(ctx, dy, S, targets):
        """the S weight gradients of this layer as ONE launch (hwg_conv_wgrad_sets / hwg_wino_wgrad_sets: the layer input is shared, every set
        has its own pixel ranges, partial images and destination buffer). -> (per-set dw list, per-set db list) with None where the kernels
        accumulated in place, or None when this layer has no grouped path (channel-padded copies, K <= 2 / C <= 2 direct kernels)"""
        global GRAD_SET
        import numpy as np
        if not ctx.needs_input_grad[1]:
            return None
        x, weight = ctx.saved_tensors
        _RUN_SCOPE[0] = ctx.scope
        stride, padding, dilation, transposed, P, Q = ctx.geom
        N, H, W, C = x.shape
        R, S_ = _taps(weight)
        sh, sw = stride; ph, pw = padding; dh, dw = dilation
        K = dy.shape[3]
        pkey = (N, H, W, C, K, R, S_, sh, sw, ph, pw, dh, dw, P, Q, transposed)
        plan = _wgrad_plans.get(pkey)
        if plan is None:
            plan = _wgrad_plans[pkey] = _make_wgrad_plan(*pkey)
        d, engine, need, Kq, Cq, tiny_end, tap_gemm = plan
        ok = _wgrad_sets_ok.get(pkey)
        if ok is None:
            ok = _wgrad_sets_ok[pkey] = engine != 1 and not (engine == 0 and transposed) and not tiny_end and (
                engine == 0 or bool(L.query("hwg_conv_wgrad_sets_supported", d.ptr)))
        if not ok:
            return None
        wref, bref = ctx.param_refs
        direct = _direct(wref)
        want_bias = ctx.has_bias and ctx.needs_input_grad[2]
        fuse_bias = want_bias and not transposed and (engine == 0 or (d.K > 2 and (d.C > 2 or tap_gemm)))
        bdirect = _direct(bref) if want_bias else False
        dws, dbs, dws_out, dbs_out = [], [], [], []
        for s_ in range(S):
            GRAD_SET = targets[s_]
            dws.append(_grad_buffer(wref) if direct else torch.empty_like(weight))
            dws_out.append(None if direct else dws[-1])
            if fuse_bias:
                dbs.append(_grad_buffer(bref) if bdirect else torch.empty((K,), dtype=torch.float32, device=x.device))
                dbs_out.append(None if bdirect else dbs[-1])
        wptr = np.array([t.data_ptr() for t in dws], dtype=np.int64)
        bptr = np.array([t.data_ptr() for t in dbs], dtype=np.int64) if fuse_bias else None
        if not transposed:
            u, v, set_on_v = dy, x, 0
            sa, sb = C * R * S_, R * S_
        else:
            u, v, set_on_v = x, dy, 1
            sa, sb = K * R * S_, R * S_
        acc = 1 if direct else 0
        bacc = 1 if (fuse_bias and bdirect) else 0
        tkey = (pkey, S)
        total = _wgrad_sets_ws.get(tkey)
        if total is None:
            total = _wgrad_sets_ws[tkey] = L.query("hwg_wino_wgrad_sets_workspace" if engine == 0 else "hwg_conv_wgrad_sets_workspace", d.ptr, S)
        can_defer = DEFER_REDUCE and direct and (not fuse_bias or bdirect) and total
        ws = _defer_workspace(total, x.device) if can_defer else None
        defer = ws is not None
        st = _stream()
        if PROF_SHAPES is not None:
            _prof_tag((d.N, d.H, d.W, d.C, d.K, R, S_, (sh, sw), (ph, pw), (dh, dw), "wgrad", ctx.scope))
        side = SIDE_WGRAD and direct and engine == 2 and (not fuse_bias or bdirect)
        if side:
            s2, raw2, skey = _side_stream(x.device)
            L.call("hwg_stream_fork", st, raw2)
            if ws is None:
                ws = _side_workspace(total, x.device, s2)
            st = raw2
            _side_hold.append((u, v))
            _side_dirty.add(skey)
        elif ws is None:
            ws = workspace(total, x.device)
        if defer:
            L.call("hwg_wgrad_defer_next")
        if engine == 0:
            L.call("hwg_wino_wgrad_sets", d.ptr, u, v, S, wptr.ctypes.data, sa, sb, S_, 1, acc, bptr.ctypes.data if fuse_bias else None, bacc,
                   ws, ws.numel(), st)
        else:
            L.call("hwg_conv_wgrad_sets", d.ptr, u, v, S, set_on_v, wptr.ctypes.data, sa, sb, S_, 1, acc, bptr.ctypes.data if fuse_bias else None, bacc,
                   ws, ws.numel(), st)
        if want_bias and not fuse_bias:
            n = dy.shape[0] // S
            for s_ in range(S):
                GRAD_SET = targets[s_]
                part = dy[s_ * n:(s_ + 1) * n]
                if bdirect:
                    colsum(part.view(-1, K), out=_grad_buffer(bref), accumulate=True)
                    dbs_out.append(None)
                else:
                    dbs_out.append(colsum(part.view(-1, K)))
        if not dbs_out:
            dbs_out = [None] * S
        return dws_out, dbs_out

    @staticmethod
    def _dgrad(ctx, dy):
        x, weight = ctx.saved_tensors
        _RUN_SCOPE[0] = ctx.scope
        stride, padding, dilation, transposed, P, Q = ctx.geom
        N, H, W, C = x.shape
        N = dy.shape[0]              # (may be a multiple of x's batch: several gradient sets through the same layer)
        R, S = _taps(weight)
        sh, sw = stride; ph, pw = padding; dh, dw = dilation
        K = dy.shape[3]
        dx = None
        st = _stream()
        if True:
            # data gradient: contraction over K (dy's channels) producing C channels
            Kp = _cpad(K, C, fractional=(not transposed and (sh != 1 or sw != 1)))
            dyin = _pad_channels(dy, Kp) if Kp != K else dy
            if not transposed and C == 1 and sh == 1 and sw == 1 and K % 16 == 0 and 1 < R * S <= 64:
                # single-channel input (first layers): dx has one channel, so instead of a matrix-vector kernel the tap matrix
                # t[pixel][tap] = dy x W^T is formed by a 1x1 convolution on the matrix cores and folded back by col2im (379 -> ~40 us for
                # the discriminator's 7x7 in_conv at 8 x 64 x 512)
                wt = _pack(weight, R * S, K, 1, 1, 1, R * S, flip=0)           # [1][taps][K]: element (tap, k) = w[k][0][tap]
                t = _run_conv(dy, wt, None, N, P, Q, K, R * S, 1, 1, (1, 1), (0, 0), (1, 1), P, Q, 0)
                dx = torch.empty((N, H, W, 1), dtype=torch.float32, device=x.device)
                L.call("hwg_col2im_taps", t, dx, N, H, W, P, Q, R, S, ph, pw, dh, dw, st)
            elif not transposed:
                if sh == 1 and sw == 1 and P == 1 and ph == 0 and dh == 1 and dw == 1 and R > 1 and Kp % 16 == 0 and C > 2:
                    # ONE row of output gradients (the recogniser's last 3x3 layer: three rows in, one out): input row r receives tap row r
                    # and nothing else - a fractionally strided layer with stride (R, 1) whose row classes have 1 x S taps each. As a stride-1
                    # correlation with padding R - 1 two thirds of the multiplies meet padding (97 -> ~55 us at 8 x 1 x 126 x 512 -> 512).
                    wp = _pack(weight, C, K, R, S, R * S, C * R * S, flip=0, Bpad=Kp)
                    dx = _run_conv(dyin, wp, None, N, P, Q, Kp, C, R, S, (R, 1), (0, pw), (1, 1), H, W, 1)
                elif sh == 1 and sw == 1:
                    wino = _wino_ok(N, P, Q, Kp, C, R, S, (1, 1), (dh * (R - 1) - ph, dw * (S - 1) - pw), (dh, dw), H, W)
                    wp = _pack(weight, C, K, R, S, R * S, C * R * S, flip=1, Bpad=Kp, wino=wino)
                    dx = _run_conv(dyin, wp, None, N, P, Q, Kp, C, R, S, (1, 1), (dh * (R - 1) - ph, dw * (S - 1) - pw), (dh, dw), H, W, 0, wino)
                elif Kp == K and _wino_s2_ok(N, P, Q, K, C, R, S, (sh, sw), (ph, pw), (dh, dw), H, W, 1):
                    wp = _pack(weight, K, C, 4, 4, C * 16, 16, flip=1, wino=2)
                    dx = _run_conv(dy, wp, None, N, P, Q, K, C, 4, 4, (2, 2), (0, 0), (1, 1), H, W, 1, 2)
                else:
                    assert dh == 1 and dw == 1, "strided conv backward needs dilation 1"
                    wp = _pack(weight, C, K, R, S, R * S, C * R * S, flip=0, Bpad=Kp)
                    # (one output row, H == R: the vertical stride never moves - stride R gives every input row its own class of 1 x S taps
                    # where stride sh pairs each class with R / sh tap rows of which all but one meet nothing; the style extractor's last
                    # 4x4 stride-(2,1) layer)
                    if P == 1 and ph == 0 and R > sh and (H == R or (H > R and ONEROW_TWIN_DEAD_ROWS)):
                        # (H > R: rows R.. of the input never met the filter - the style extractor's last block sees 5 rows and uses 4. They
                        # get zeros; the R live rows run as the stride-R twin instead of sh-strided classes whose second block row is dead:
                        # 4x1x254x256 -> 4x5x257x256: 81 -> ~50 us)
                        dx = _run_conv(dyin, wp, None, N, P, Q, Kp, C, R, S, (R, sw), (ph, pw), (1, 1), R, W, 1)
                        if H > R:
                            full = torch.empty((N, H, W, C), dtype=torch.float32, device=dx.device)
                            L.call("hwg_pad2d_fwd", dx, full, N, R, W, C, 0, H - R, 0, 0, 0, 0.0, st)
                            dx = full
                    else:
                        dx = _run_conv(dyin, wp, None, N, P, Q, Kp, C, R, S, (sh, sw), (ph, pw), (1, 1), H, W, 1)
            else:
                # gradient of a transposed conv is an ordinary (strided) correlation of dy
                wino = _wino_ok(N, P, Q, Kp, C, R, S, (sh, sw), (ph, pw), (dh, dw), H, W)
                wp = _pack(weight, C, K, R, S, K * R * S, R * S, flip=0, Bpad=Kp, wino=wino)
                dx = _run_conv(dyin, wp, None, N, P, Q, Kp, C, R, S, (sh, sw), (ph, pw), (dh, dw), H, W, 0, wino)
        return dx

    @staticmethod
    def _wgrad(ctx, dy):
        """-> (dw, db) as tensors, or None where the kernel accumulated straight into the parameter's gradient buffer"""
        x, weight = ctx.saved_tensors
        _RUN_SCOPE[0] = ctx.scope
        stride, padding, dilation, transposed, P, Q = ctx.geom
        N, H, W, C = x.shape
        R, S = _taps(weight)
        sh, sw = stride; ph, pw = padding; dh, dw = dilation
        K = dy.shape[3]
        dw_ = db = None
        bias_done = False
        st = _stream()
        wref, bref = ctx.param_refs
        if ctx.needs_input_grad[1]:
            direct = _direct(wref)
            dw_ = _grad_buffer(wref) if direct else torch.empty_like(weight)
            # descriptor, engine choice and workspace size are fixed per geometry: built once (the library's cost models run there)
            pkey = (N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, P, Q, transposed)
            plan = _wgrad_plans.get(pkey)
            if plan is None:
                plan = _wgrad_plans[pkey] = _make_wgrad_plan(*pkey)
            d, engine, need, Kq, Cq, tiny_end, tap_gemm = plan
            if not transposed:
                u, v = dy, x
                sa, sb = C * R * S, R * S
            else:
                u, v = x, dy
                sa, sb = K * R * S, R * S
            # the sum of the partial images may wait for the flush behind the backward pass where nothing reads the gradient earlier: a
            # parameter's buffer, or the dW_sn of a spectral-norm layer whose own backward is queued for the same flush (_SpectralScale)
            late = direct or getattr(wref, "_hwg_sn_defer", False)
            dws = _defer_workspace(need, x.device) if (DEFER_REDUCE and late and engine != 1 and need) else None
            if engine == 0:
                ws = dws if dws is not None else workspace(need, x.device)
                if PROF_SHAPES is not None:
                    _prof_tag((d.N, d.H, d.W, d.C, d.K, R, S, (sh, sw), (ph, pw), (dh, dw), "wgrad", ctx.scope))
                dbias = bacc = None
                if ctx.has_bias and ctx.needs_input_grad[2] and not transposed:     # dy is the kernel's anchor operand: its column sums ride along
                    bdirect = _direct(bref)
                    dbias = _grad_buffer(bref) if bdirect else torch.empty((K,), dtype=torch.float32, device=x.device)
                    bacc = 1 if bdirect else 0
                    bias_done = True
                    if not bdirect:
                        db = dbias
                if dws is not None and (dbias is None or bacc):
                    L.call("hwg_wgrad_defer_next")
                L.call("hwg_wino_wgrad", d.ptr, u, v, dw_, sa, sb, S, 1, 1 if direct else 0, dbias, bacc or 0, ws, ws.numel(), st)
                if direct:
                    dw_ = None
            elif engine == 1:
                # channel counts that are not multiples of 4 (RIMES: 78 classes -> 206/334-channel inputs, 78 outputs): run the kernel
                # on zero-padded copies and keep the valid block of the result (the plan's descriptor carries the padded counts)
                dK, dC = (K, C) if not transposed else (C, K)
                up = _pad_channels(u, Kq) if Kq != dK else u
                vp = _pad_channels(v, Cq) if Cq != dC else v
                tmp = torch.empty((Kq, Cq, R, S), dtype=torch.float32, device=x.device)
                ws = workspace(need, x.device)
                if PROF_SHAPES is not None:
                    _prof_tag((d.N, d.H, d.W, d.C, d.K, R, S, (sh, sw), (ph, pw), (dh, dw), "wgrad", ctx.scope))
                L.call("hwg_conv_wgrad", d.ptr, up, vp, tmp, Cq * R * S, R * S, S, 1, 0, None, 0, ws, ws.numel(), st)
                # the valid block of the padded result, added to (or copied into) the gradient through the C-ABI: row k of `tmp` holds Cq*R*S floats
                # of which the first dC*R*S belong to real channels (a torch-side add_ here was invisible to a recorded call list: every RIMES
                # geometry of the recogniser failed its replay self-check)
                L.call("hwg_copy_channels", tmp, Cq * R * S, 0, dw_, dC * R * S, 0, dC * R * S, dK, 1, 0, 1 if direct else 0, st)
                if direct:
                    dw_ = None
            else:
                ws = dws if dws is not None else workspace(need, x.device)
                if PROF_SHAPES is not None:
                    _prof_tag((d.N, d.H, d.W, d.C, d.K, R, S, (sh, sw), (ph, pw), (dh, dw), "wgrad", ctx.scope))
                # the bias gradient (column sums of dy) rides along when dy is the kernel's anchor operand and the MFMA path runs
                fuse_bias = (ctx.has_bias and ctx.needs_input_grad[2] and not transposed and not tiny_end and d.K > 2 and (d.C > 2 or tap_gemm))
                dbias = bacc = None
                if fuse_bias:
                    bdirect = _direct(bref)
                    dbias = _grad_buffer(bref) if bdirect else torch.empty((K,), dtype=torch.float32, device=x.device)
                    bacc = 1 if bdirect else 0
                    bias_done = True
                    if not bdirect:
                        db = dbias
                defer = dws is not None and (dbias is None or bacc)
                if SIDE_WGRAD and direct and (dbias is None or bacc):
                    s2, raw2, skey = _side_stream(x.device)
                    L.call("hwg_stream_fork", st, raw2)      # dy (and x) are complete on the main stream at this point
                    ws2 = dws if defer else _side_workspace(need, x.device, s2)
                    if defer:
                        L.call("hwg_wgrad_defer_next")
                    L.call("hwg_conv_wgrad", d.ptr, u, v, dw_, sa, sb, S, 1, 1, dbias, bacc or 0, ws2, ws2.numel(), raw2)
                    _side_hold.append((u, v))
                    _side_dirty.add(skey)
                else:
                    if defer:
                        L.call("hwg_wgrad_defer_next")
                    L.call("hwg_conv_wgrad", d.ptr, u, v, dw_, sa, sb, S, 1, 1 if direct else 0, dbias, bacc or 0, ws, ws.numel(), st)
                if direct:
                    dw_ = None
        if ctx.has_bias and ctx.needs_input_grad[2] and not bias_done:
            if _direct(bref):
                colsum(dy.view(-1, K), out=_grad_buffer(bref), accumulate=True)
            else:
                db = colsum(dy.view(-1, K))
        return dw_, db


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1):
    return _Conv2d.apply(x, weight, bias, _pair(stride), _pair(padding), _pair(dilation), False, (0, 0))


def conv_transpose2d(x, weight, bias=None, stride=1, padding=0, output_padding=0, dilation=1):
    return _Conv2d.apply(x, weight, bias, _pair(stride), _pair(padding), _pair(dilation), True, _pair(output_padding))


def conv1d(x, weight, bias=None, stride=1, padding=0, dilation=1):
    """x [N,1,L,C]; weight in Conv1d layout [O,I,S]"""
    return conv2d(x, weight, bias, (1, stride), (0, padding), (1, dilation))


def linear(x, weight, bias=None):
    """x [rows, I] -> [rows, O]; weight [O, I]"""
    rows, I = x.shape
    y = conv2d(x.view(rows, 1, 1, I), weight, bias)
    return y.view(rows, weight.shape[0])


def colsum(x2d, out=None, accumulate=False):
    rows, C = x2d.shape
    if out is None:
        out = torch.empty((C,), dtype=torch.float32, device=x2d.device)
    need = L.query("hwg_colsum_workspace", rows, C)
    ws = workspace(need, x2d.device)
    L.call("hwg_colsum", x2d, rows, C, out, int(accumulate), ws, ws.numel(), _stream())
    return out


# ----------------------------------------------------------------------------------------------
# normalisation + activation
# ----------------------------------------------------------------------------------------------
def _affine_stamp(gamma, beta):
    """(version counters, optimizer epoch) of a norm's affine parameters: HipAdam writes through raw pointers behind torch's version
    counters and bumps WEIGHT_EPOCH of the parameter's group instead"""
    return tuple((p._version, WEIGHT_EPOCH.get(getattr(p, "_hwg_group", None), 0)) for p in (gamma, beta) if p is not None)


class _Norm(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, mode, groups, eps, mask, act, slope, running_mean, running_var, momentum):
        _chk(x, "norm input"); _chk(gamma, "norm weight"); _chk(beta, "norm bias"); _chk(mask, "channel mask")
        N, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (N * C)
        y = torch.empty_like(x)
        mean = torch.empty((N, C), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = workspace(L.query("hwg_norm_workspace", N, HW, C), x.device)
        L.call("hwg_norm_fwd", x, y, N, HW, C, mode, groups, eps, gamma, beta, 0, mask, act, slope, mean, rstd,
               running_mean, running_var, momentum, ws, ws.numel(), _stream())
        ctx.save_for_backward(x, y, gamma, mask, mean, rstd)
        ctx.cfg = (mode, groups, act, slope, N, HW, C, beta is not None)
        ctx.param_refs = (gamma, beta)
        # the backward pass recomputes relu / leaky-relu gates from x and the affine it reads THEN: remember which weights the forward saw
        ctx.wstamp = _affine_stamp(gamma, beta) if act in (ACT_RELU, ACT_LRELU) else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mask, mean, rstd = ctx.saved_tensors
        mode, groups, act, slope, N, HW, C, has_beta = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        gref, bref = ctx.param_refs
        if ctx.wstamp is not None and ctx.wstamp != _affine_stamp(gref, bref):
            raise L.HwgError("norm backward after its affine parameters changed (optimizer step / in-place write between forward and backward): the "
                             "activation gates are recomputed from x and the current affine and would be wrong; run backward before stepping")
        direct = (gamma is None or _direct(gref)) and (not has_beta or _direct(bref))
        if direct:
            dgamma = _grad_buffer(gref) if gamma is not None else None
            dbeta = _grad_buffer(bref) if has_beta else None
        else:
            dgamma = torch.empty((C,), dtype=torch.float32, device=x.device) if gamma is not None else None
            dbeta = torch.empty((C,), dtype=torch.float32, device=x.device) if has_beta else None
        ws = workspace(L.query("hwg_norm_workspace", N, HW, C), x.device)
        # beta: relu / leaky relu gates are recomputed from x and the forward call's affine instead of being read from y (csrc/norm_act.hip)
        L.call("hwg_norm_bwd", dy, x, y, dx, N, HW, C, mode, groups, gamma, bref if has_beta else None, 0, mask, act, slope, mean, rstd, dgamma, dbeta, 1 if direct else 0,
               ws, ws.numel(), _stream())
        if direct:
            dgamma = dbeta = None
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None, None


def group_norm(x, groups, gamma, beta, eps=1e-5, mask=None, act=ACT_NONE, slope=0.0):
    return _Norm.apply(x, gamma, beta, NORM_GN, groups, eps, mask, act, slope, None, None, 0.0)


def batch_norm_train(x, gamma, beta, running_mean, running_var, momentum=0.1, eps=1e-5, act=ACT_NONE, slope=0.0):
    return _Norm.apply(x, gamma, beta, NORM_BN, 1, eps, None, act, slope, running_mean, running_var, momentum)


def instance_norm(x, eps=1e-5):
    return _Norm.apply(x, None, None, NORM_IN, 1, eps, None, ACT_NONE, 0.0, None, None, 0.0)


class _AdaIN(Function):
    """(x, noise, noise_weight_orig, gamma, beta) -> gamma*IN(lrelu(x + w*noise)) + beta  (+ folded conv-bias gradient)"""

    @staticmethod
    def forward(ctx, x, noise, noise_w, gamma, beta, noise_scale, slope, eps):
        for t, n in ((x, "x"), (noise, "noise"), (noise_w, "noise weight"), (gamma, "gamma"), (beta, "beta")):
            _chk(t, "adain " + n)
        N, C = x.shape[0], x.shape[-1]
        HW = x.numel() // (N * C)
        u = torch.empty_like(x)
        y = torch.empty_like(x)
        mean = torch.empty((N, C), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = workspace(L.query("hwg_norm_workspace", N, HW, C), x.device)
        L.call("hwg_adain_fwd", x, noise, noise_w, noise_scale, slope, gamma, beta, eps, u, y, mean, rstd, N, HW, C, ws, ws.numel(), _stream())
        ctx.save_for_backward(u, noise, gamma, mean, rstd)
        ctx.cfg = (noise_scale, slope, N, HW, C, noise_w.shape)
        ctx.param_refs = (noise_w,)
        return y

    @staticmethod
    def backward(ctx, dy):
        u, noise, gamma, mean, rstd = ctx.saved_tensors
        noise_scale, slope, N, HW, C, wshape = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(u)
        dgamma = torch.empty((N, C), dtype=torch.float32, device=u.device)
        dbeta = torch.empty_like(dgamma)
        (nwref,) = ctx.param_refs
        direct = _direct(nwref)
        dnw = _grad_buffer(nwref) if direct else torch.empty((C,), dtype=torch.float32, device=u.device)
        need = L.query("hwg_norm_workspace", N, HW, C)
        ws = _defer_workspace(need, u.device) if (DEFER_REDUCE and direct) else None      # the noise-weight sum joins the pass's one deferred launch
        if ws is not None:
            L.call("hwg_wgrad_defer_next")
        else:
            ws = workspace(need, u.device)
        L.call("hwg_adain_bwd", dy, u, noise, noise_scale, slope, gamma, mean, rstd, dx, dgamma, dbeta, dnw, None, 1 if direct else 0, N, HW, C,
               ws, ws.numel(), _stream())
        return dx, None, (None if direct else dnw.view(wshape)), dgamma, dbeta, None, None, None

    sets = "native"

    @staticmethod
    def backward_sets(ctx, S, targets, dy):
        """S gradient sets stacked along the batch axis: the saved activations are shared, so the kernel runs once per set on that set's slice
        and writes its slice of the stacked results (no concatenation pass); the noise-weight gradient of set s goes to set s's buffer"""
        global GRAD_SET
        u, noise, gamma, mean, rstd = ctx.saved_tensors
        noise_scale, slope, N, HW, C, wshape = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty((S * N,) + tuple(u.shape[1:]), dtype=torch.float32, device=u.device)
        dgamma = torch.empty((S * N, C), dtype=torch.float32, device=u.device)
        dbeta = torch.empty_like(dgamma)
        (nwref,) = ctx.param_refs
        if not _direct(nwref):
            raise L.HwgError("batched AdaIN backward needs the noise weight's gradient buffer (a leaf parameter)")
        need = L.query("hwg_norm_workspace", N, HW, C)
        st = _stream()
        for s_ in range(S):
            GRAD_SET = targets[s_]
            dnw = _grad_buffer(nwref)
            sl = slice(s_ * N, (s_ + 1) * N)
            ws = _defer_workspace(need, u.device) if DEFER_REDUCE else None
            if ws is not None:
                L.call("hwg_wgrad_defer_next")
            else:
                ws = workspace(need, u.device)
            L.call("hwg_adain_bwd", dy[sl], u, noise, noise_scale, slope, gamma, mean, rstd, dx[sl], dgamma[sl], dbeta[sl], dnw, None, 1, N, HW, C,
                   ws, ws.numel(), st)
        return dx, None, None, dgamma, dbeta, None, None, None


class VirtualNoise:
    """a noise tensor that is never written: `shape` standard normals = what hwg_randn(seed, offset) would put into a tensor of that shape
    (rng.NoiseBlock hands these out in forward-only passes; adain_epilogue draws the values inside its kernel)"""

    __slots__ = ("seed", "offset", "shape")

    def __init__(self, seed, offset, shape):
        self.seed, self.offset, self.shape = int(seed), int(offset), tuple(shape)

    def materialise(self, device):
        out = torch.empty(self.shape, dtype=torch.float32, device=device)
        L.call("hwg_randn", out, out.numel(), self.seed, self.offset, _stream())
        return out


def adain_epilogue(x, noise, noise_w, gamma, beta, noise_scale, slope=0.2, eps=1e-5):
    if isinstance(noise, VirtualNoise):
        if TAPE is not None or (torch.is_grad_enabled() and any(t.requires_grad for t in (x, noise_w, gamma, beta))):
            noise = noise.materialise(x.device)          # (a backward pass will read the noise)
        else:
            for t, n in ((x, "x"), (noise_w, "noise weight"), (gamma, "gamma"), (beta, "beta")):
                _chk(t, "adain " + n)
            assert tuple(x.shape) == noise.shape
            N, C = x.shape[0], x.shape[-1]
            HW = x.numel() // (N * C)
            u = torch.empty_like(x)
            y = torch.empty_like(x)
            mean = torch.empty((N, C), dtype=torch.float32, device=x.device)
            rstd = torch.empty_like(mean)
            ws = workspace(L.query("hwg_norm_workspace", N, HW, C), x.device)
            L.call("hwg_adain_fwd_rng", x, noise.seed, noise.offset, noise_w, noise_scale, slope, gamma, beta, eps, u, y, mean, rstd, N, HW, C,
                   ws, ws.numel(), _stream())
            return y
    return _AdaIN.apply(x, noise, noise_w, gamma, beta, noise_scale, slope, eps)


class _BiasAct(Function):
    @staticmethod
    def forward(ctx, x, bias, mask, act, slope):
        _chk(x, "bias_act input"); _chk(bias, "bias"); _chk(mask, "channel mask")
        C = x.shape[-1]
        N = x.shape[0]
        rows = x.numel() // C
        HW = rows // N
        y = torch.empty_like(x)
        L.call("hwg_bias_act_fwd", x, bias, mask, y, rows, HW, C, act, slope, _stream())
        ctx.save_for_backward(y, mask)
        ctx.cfg = (act, slope, rows, HW, C, bias is not None)
        ctx.param_refs = (bias,)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, mask = ctx.saved_tensors
        act, slope, rows, HW, C, has_bias = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        L.call("hwg_bias_act_bwd", dy, y, mask, dx, rows, HW, C, act, slope, _stream())
        db = None
        if has_bias and ctx.needs_input_grad[1]:
            (bref,) = ctx.param_refs
            if _direct(bref):
                colsum(dx.view(rows, C), out=_grad_buffer(bref), accumulate=True)
            else:
                db = colsum(dx.view(rows, C))
        return dx, db, None, None, None


def bias_act(x, bias=None, mask=None, act=ACT_NONE, slope=0.0):
    return _BiasAct.apply(x, bias, mask, act, slope)


def relu(x):
    return _BiasAct.apply(x, None, None, ACT_RELU, 0.0)


def leaky_relu(x, slope):
    return _BiasAct.apply(x, None, None, ACT_LRELU, slope)


class _Tanh(Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x, "tanh input")
        y = torch.empty_like(x)
        L.call("hwg_tanh_fwd", x, y, x.numel(), _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        L.call("hwg_tanh_bwd", dy.contiguous(), y, dx, y.numel(), _stream())
        return dx


def tanh(x):
    return _Tanh.apply(x)


class _PixelNorm(Function):
    @staticmethod
    def forward(ctx, x, eps):
        _chk(x, "pixelnorm input")
        rows, C = x.shape
        y = torch.empty_like(x)
        L.call("hwg_pixelnorm_fwd", x, y, rows, C, eps, _stream())
        ctx.save_for_backward(x)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        L.call("hwg_pixelnorm_bwd", dy.contiguous(), x, dx, x.shape[0], x.shape[1], ctx.eps, _stream())
        return dx, None


def pixel_norm(x, eps=1e-8):
    return _PixelNorm.apply(x, eps)


# ----------------------------------------------------------------------------------------------
# pooling / resampling / padding / concat
# ----------------------------------------------------------------------------------------------
class _AvgPool(Function):
    sets = "stacked"

    @staticmethod
    def forward(ctx, x, kh, kw):
        _chk(x, "avgpool input")
        N, H, W, C = x.shape
        y = torch.empty((N, H // kh, W // kw, C), dtype=torch.float32, device=x.device)
        L.call("hwg_avgpool_fwd", x, y, N, H, W, C, kh, kw, _stream())
        ctx.cfg = (N, H, W, C, kh, kw)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C, kh, kw = ctx.cfg
        N = dy.shape[0]           # (several gradient sets stacked along the batch axis: Tape.backward_sets)
        dx = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
        L.call("hwg_avgpool_bwd", dy.contiguous(), dx, N, H, W, C, kh, kw, _stream())
        return dx, None, None


def avg_pool2d(x, kernel):
    kh, kw = _pair(kernel)
    return _AvgPool.apply(x, kh, kw)


class _ActAvgPool(Function):
    """avg_pool2d(act(mask * x)) as one kernel per direction (hwg_act_avgpool_*): bit-identical to bias_act + avg_pool2d, without writing and
    re-reading the full-resolution activation (forward) and its gradient (backward); the gate is recomputed from x"""

    @staticmethod
    def forward(ctx, x, mask, act, slope, kh, kw):
        _chk(x, "act_avgpool input"); _chk(mask, "channel mask")
        N, H, W, C = x.shape
        y = torch.empty((N, H // kh, W // kw, C), dtype=torch.float32, device=x.device)
        L.call("hwg_act_avgpool_fwd", x, mask, y, N, H, W, C, kh, kw, act, slope, _stream())
        ctx.save_for_backward(x, mask)
        ctx.cfg = (N, H, W, C, kh, kw, act, slope)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mask = ctx.saved_tensors
        N, H, W, C, kh, kw, act, slope = ctx.cfg
        dx = torch.empty_like(x)
        L.call("hwg_act_avgpool_bwd", dy.contiguous(), x, mask, dx, N, H, W, C, kh, kw, act, slope, _stream())
        return dx, None, None, None, None, None


def act_avg_pool2d(x, kernel, mask=None, act=ACT_NONE, slope=0.0):
    """avg_pool2d(bias_act(x, None, mask, act, slope), kernel) in one pass"""
    kh, kw = _pair(kernel)
    return _ActAvgPool.apply(x, mask, act, slope, kh, kw)


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x, kernel, stride, padding, relu):
        _chk(x, "maxpool input")
        N, H, W, C = x.shape
        kh, kw = kernel; sh, sw = stride; ph, pw = padding
        P = (H + 2 * ph - kh) // sh + 1
        Q = (W + 2 * pw - kw) // sw + 1
        y = torch.empty((N, P, Q, C), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, P, Q, C), dtype=torch.int32, device=x.device)
        L.call("hwg_maxpool_relu_fwd" if relu else "hwg_maxpool_fwd", x, y, idx, N, H, W, C, kh, kw, sh, sw, ph, pw, P, Q, _stream())
        if relu:
            ctx.save_for_backward(idx, y)
        else:
            ctx.save_for_backward(idx)
        ctx.cfg = (N, H, W, C, kh, kw, sh, sw, ph, pw, P, Q, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C, kh, kw, sh, sw, ph, pw, P, Q, relu = ctx.cfg
        dx = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
        if relu:
            idx, y = ctx.saved_tensors
            L.call("hwg_maxpool_relu_bwd", dy.contiguous(), y, idx, dx, N, H, W, C, kh, kw, sh, sw, ph, pw, P, Q, _stream())
        else:
            (idx,) = ctx.saved_tensors
            L.call("hwg_maxpool_bwd", dy.contiguous(), idx, dx, N, H, W, C, kh, kw, sh, sw, ph, pw, P, Q, _stream())
        return dx, None, None, None, None


def max_pool2d(x, kernel, stride=None, padding=0, relu=False):
    """relu=True: relu(max_pool2d(x)) (== max_pool2d(relu(x)) exactly) with the ReLU riding along in the pooling kernels, both directions"""
    kernel = _pair(kernel)
    stride = kernel if stride is None else _pair(stride)
    return _MaxPool.apply(x, kernel, stride, _pair(padding), bool(relu))


class _Upsample(Function):
    sets = "stacked"

    @staticmethod
    def forward(ctx, x, fh, fw):
        _chk(x, "upsample input")
        N, H, W, C = x.shape
        y = torch.empty((N, H * fh, W * fw, C), dtype=torch.float32, device=x.device)
        L.call("hwg_upsample_nearest_fwd", x, y, N, H, W, C, fh, fw, _stream())
        ctx.cfg = (N, H, W, C, fh, fw)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C, fh, fw = ctx.cfg
        N = dy.shape[0]
        dx = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
        L.call("hwg_upsample_nearest_bwd", dy.contiguous(), dx, N, H, W, C, fh, fw, _stream())
        return dx, None, None


def upsample_nearest(x, scale):
    fh, fw = _pair(scale)
    return _Upsample.apply(x, fh, fw)


class _Blur(Function):
    sets = "stacked"

    @staticmethod
    def forward(ctx, x):
        _chk(x, "blur input")
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        L.call("hwg_blur3", x, y, N, H, W, C, _stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        return _Blur.apply(dy.contiguous())  # symmetric kernel; differentiable again like the reference's BlurFunctionBackward


def blur3(x):
    return _Blur.apply(x)


class _Pad2d(Function):
    sets = "stacked"

    @staticmethod
    def forward(ctx, x, pt, pb, pl, pr, mode, value):
        _chk(x, "pad input")
        N, H, W, C = x.shape
        y = torch.empty((N, H + pt + pb, W + pl + pr, C), dtype=torch.float32, device=x.device)
        L.call("hwg_pad2d_fwd", x, y, N, H, W, C, pt, pb, pl, pr, mode, value, _stream())
        ctx.cfg = (N, H, W, C, pt, pb, pl, pr, mode)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C, pt, pb, pl, pr, mode = ctx.cfg
        N = dy.shape[0]
        dx = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
        L.call("hwg_pad2d_bwd", dy.contiguous(), dx, N, H, W, C, pt, pb, pl, pr, mode, _stream())
        return dx, None, None, None, None, None, None


def pad2d(x, left, right, top=0, bottom=0, mode="constant", value=0.0):
    if left == 0 and right == 0 and top == 0 and bottom == 0:
        return x
    return _Pad2d.apply(x, top, bottom, left, right, 1 if mode == "replicate" else 0, float(value))


class _CatChannels(Function):
    """concatenate along C; parts that are 2-D [N, c] are broadcast over the pixels of sample n"""
    sets = "stacked"

    @staticmethod
    def forward(ctx, ref_shape, *parts):
        N, H, W = ref_shape
        rows = N * H * W
        widths = [p.shape[-1] for p in parts]
        Ct = sum(widths)
        out = torch.empty((N, H, W, Ct), dtype=torch.float32, device=parts[0].device)
        off = 0
        st = _stream()
        for p, c in zip(parts, widths):
            _chk(p, "cat part")
            bcast = 1 if p.dim() == 2 else 0
            L.call("hwg_copy_channels", p, c, 0, out, Ct, off, c, rows, H * W, bcast, 0, st)
            off += c
        ctx.cfg = (N, H, W, widths, [p.dim() == 2 for p in parts])
        return out

    @staticmethod
    def backward(ctx, dy):
        N, H, W, widths, bc = ctx.cfg
        dy = dy.contiguous()
        N = dy.shape[0]
        Ct = sum(widths)
        rows = N * H * W
        grads = []
        off = 0
        st = _stream()
        for i, (c, b) in enumerate(zip(widths, bc)):
            if not ctx.needs_input_grad[1 + i]:
                grads.append(None)
            elif b:
                g = torch.empty((N, c), dtype=torch.float32, device=dy.device)
                L.call("hwg_reduce_rows", dy, Ct, off, g, c, N, H * W, 0, st)
                grads.append(g)
            else:
                g = torch.empty((N, H, W, c), dtype=torch.float32, device=dy.device)
                L.call("hwg_copy_channels", dy, Ct, off, g, c, 0, c, rows, H * W, 0, 0, st)
                grads.append(g)
            off += c
        return (None,) + tuple(grads)


def cat_channels(parts, spatial):
    """parts: list of [N,H,W,c] or [N,c] (broadcast) tensors; spatial = (N,H,W)"""
    return _CatChannels.apply(tuple(spatial), *parts)


def onehot_rows(label_LB, ncls):
    """label int32 [L,B] (device) -> float [B,1,L,ncls]"""
    Lr, B = label_LB.shape
    _chk(label_LB, "label", torch.int32)
    out = torch.empty((B, 1, Lr, ncls), dtype=torch.float32, device=label_LB.device)
    L.call("hwg_onehot", label_LB, out, Lr, B, ncls, ncls, 0, _stream())
    return out


def onehot_both(label_LB, ncls):
    """label int32 [L,B] (device) -> (time-major [L,B,ncls] with the NHWC rows [B,1,L,ncls] hanging on it as `_hwg_nhwc`), one launch: the
    consumers that read NHWC (generator, spacer) take the attached tensor instead of permuting the time-major one back"""
    Lr, B = label_LB.shape
    _chk(label_LB, "label", torch.int32)
    blc = torch.empty((B, 1, Lr, ncls), dtype=torch.float32, device=label_LB.device)
    lbc = torch.empty((Lr, B, ncls), dtype=torch.float32, device=label_LB.device)
    L.call("hwg_onehot_both", label_LB, blc, lbc, Lr, B, ncls, _stream())
    lbc._hwg_nhwc = blc
    return lbc


def nhwc_of(content_LBC):
    """the NHWC twin of a time-major tensor made by onehot_both (None for any other tensor, e.g. a slice of it)"""
    twin = getattr(content_LBC, "_hwg_nhwc", None)
    if twin is not None and content_LBC.dim() == 3 and twin.shape[0] == content_LBC.shape[1] and twin.shape[2] == content_LBC.shape[0]:
        return twin
    return None


def permute4(x, dims, strides):
    out = torch.empty(tuple(dims), dtype=torch.float32, device=x.device)
    L.call("hwg_permute4", x, out, dims[0], dims[1], dims[2], dims[3], strides[0], strides[1], strides[2], strides[3], _stream())
    return out


class _ToNCHW(Function):
    """layout change at the module boundary (NHWC -> NCHW copy); only used for multi-channel boundary tensors"""
    sets = "stacked"

    @staticmethod
    def forward(ctx, x):
        N, H, W, C = x.shape
        return permute4(x, (N, C, H, W), (H * W * C, 1, W * C, C))

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W = dy.shape
        return permute4(dy.contiguous(), (N, H, W, C), (C * H * W, W, 1, H * W))


class _ToNHWC(Function):
    sets = "stacked"

    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        return permute4(x, (N, H, W, C), (C * H * W, W, 1, H * W))

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C = dy.shape
        return permute4(dy.contiguous(), (N, C, H, W), (H * W * C, 1, W * C, C))


def to_nchw(x):
    if x.shape[3] == 1:
        y = x.reshape(x.shape[0], 1, x.shape[1], x.shape[2])
        return TAPE.alias(y, x) if TAPE is not None else y
    return _ToNCHW.apply(x)


def to_nhwc(x):
    if x.shape[1] == 1:
        y = x.reshape(x.shape[0], x.shape[2], x.shape[3], 1)
        return TAPE.alias(y, x) if TAPE is not None else y
    return _ToNHWC.apply(x.contiguous())


# ----------------------------------------------------------------------------------------------
# sequence ops
# ----------------------------------------------------------------------------------------------
class _LogSoftmaxTB(Function):
    """x [B,1,T,C] (NHWC) -> log-probs [T,B,C]"""

    @staticmethod
    def forward(ctx, x):
        _chk(x, "log_softmax input")
        B, _, T, C = x.shape
        y = torch.empty((T, B, C), dtype=torch.float32, device=x.device)
        L.call("hwg_log_softmax_fwd", x, y, B * T, C, B, T, 1, _stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        T, B, C = y.shape
        dx = torch.empty((B, 1, T, C), dtype=torch.float32, device=y.device)
        L.call("hwg_log_softmax_bwd", dy.contiguous(), y, dx, B * T, C, B, T, 1, _stream())
        return dx


def log_softmax_tbc(x):
    return _LogSoftmaxTB.apply(x)


class _CTC(Function):
    @staticmethod
    def forward(ctx, log_probs, targets, in_len, tg_len):
        _chk(log_probs, "ctc log_probs")
        for t, n in ((targets, "targets"), (in_len, "input lengths"), (tg_len, "target lengths")):
            _chk(t, "ctc " + n, torch.int32)
        T, B, C = log_probs.shape
        Lmax = targets.shape[1]
        nbytes = L.query("hwg_ctc_workspace", T, B, Lmax)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=log_probs.device)  # kept for backward
        loss = torch.empty((), dtype=torch.float32, device=log_probs.device)
        L.call("hwg_ctc_fwd", log_probs, targets, in_len, tg_len, T, B, C, Lmax, loss, ws, ws.numel(), _stream())
        ctx.save_for_backward(log_probs, targets, in_len, tg_len, ws)
        return loss

    @staticmethod
    def backward(ctx, gout):
        log_probs, targets, in_len, tg_len, ws = ctx.saved_tensors
        T, B, C = log_probs.shape
        grad = torch.empty_like(log_probs)
        L.call("hwg_ctc_bwd", log_probs, targets, in_len, tg_len, T, B, C, targets.shape[1], gout.contiguous(), grad, ws, ws.numel(), _stream())
        return grad, None, None, None


def ctc_loss(log_probs, targets, input_lengths, target_lengths):
    """F.ctc_loss(blank=0, reduction='mean') with an infinite result reported as 0 (model/loss.py:28-30).
    targets [B, Lmax]; lengths may be host tensors / lists (they are host data in the reference too)."""
    dev = log_probs.device
    targets = h2d(targets, dev, torch.int32).contiguous()
    il = h2d(torch.as_tensor(input_lengths, dtype=torch.int32), dev)
    tl = h2d(torch.as_tensor(target_lengths, dtype=torch.int32), dev)
    return _CTC.apply(log_probs.contiguous(), targets, il, tl)


class DTWPending:
    """alignment kernel already enqueued; `.result()` waits (side-stream copy) only for the path lengths"""

    def __init__(self, out, lens):
        self.out, self.lens = out, lens
        self.fetch = AsyncFetch(lens)

    def result(self):
        maxlen = int(self.fetch.get().max())
        return self.out[:maxlen].contiguous(), self.lens


def dtw_align_async(pred_TBC, label_LB):
    _chk(pred_TBC, "dtw pred")
    T, B, C = pred_TBC.shape
    label = h2d(label_LB, pred_TBC.device, torch.int32).contiguous()
    Lr = label.shape[0]
    out = torch.empty((T + 2 * Lr + 1, B), dtype=torch.int64, device=pred_TBC.device)
    lens = torch.empty((B,), dtype=torch.int32, device=pred_TBC.device)
    ws = torch.empty(L.query("hwg_dtw_workspace", T, B, Lr), dtype=torch.uint8, device=pred_TBC.device)   # private: outlives this call
    L.call("hwg_dtw_align", pred_TBC, label, T, B, C, Lr, out, lens, ws, ws.numel(), _stream())
    return DTWPending(out, lens)


def dtw_align(pred_TBC, label_LB):
    """correct_pred: returns (aligned int64 [maxlen,B], lens[B]); one D2H wait for the path length."""
    return dtw_align_async(pred_TBC, label_LB).result()


def gt_counts(index_spaced, label_LB):
    """run-length scan of an aligned label sequence -> (gt_counts [L,B,2], min pos)"""
    Tp, B = index_spaced.shape
    label = h2d(label_LB, index_spaced.device, torch.int32).contiguous()
    Lr = label.shape[0]
    gt = torch.zeros((Lr, B, 2), dtype=torch.float32, device=index_spaced.device)
    meta = h2d(torch.tensor([2 ** 31 - 1, 0], dtype=torch.int32), index_spaced.device)
    L.call("hwg_gt_counts", index_spaced.contiguous(), label, Tp, B, Lr, gt, meta[0:1], meta[1:2], _stream())
    return gt, meta


def argmax_rows(x2d):
    rows, C = x2d.shape
    out = torch.empty((rows,), dtype=torch.int32, device=x2d.device)
    L.call("hwg_argmax_rows", x2d, out, rows, C, _stream())
    return out


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
LOSS_L1, LOSS_MSE, LOSS_MEAN, LOSS_HINGE_REAL, LOSS_HINGE_FAKE = 0, 1, 2, 3, 4


class _Loss(Function):
    @staticmethod
    def forward(ctx, a, b, mode, scale):
        _chk(a, "loss input"); _chk(b, "loss target")
        out = torch.empty((), dtype=torch.float32, device=a.device)
        ws = workspace(L.query("hwg_loss_workspace"), a.device)
        L.call("hwg_loss_fwd", a, b, a.numel(), mode, scale, out, 0, ws, ws.numel(), _stream())
        ctx.save_for_backward(a, b)
        ctx.cfg = (mode, scale)
        return out

    @staticmethod
    def backward(ctx, gout):
        a, b = ctx.saved_tensors
        mode, scale = ctx.cfg
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if (b is not None and ctx.needs_input_grad[1]) else None
        if da is None and db is None:
            return None, None, None, None
        L.call("hwg_loss_bwd", a, b, a.numel(), mode, scale, gout.contiguous(), da, db, 0, _stream())
        return da, db, None, None


def l1_loss(a, b):
    return _Loss.apply(a.contiguous(), b.contiguous(), LOSS_L1, 1.0)


def mse_loss(a, b):
    return _Loss.apply(a.contiguous(), b.contiguous(), LOSS_MSE, 1.0)


def mean_loss(a, mode=LOSS_MEAN, scale=1.0):
    return _Loss.apply(a.contiguous(), None, mode, scale)


# ----------------------------------------------------------------------------------------------
# spectral norm
# ----------------------------------------------------------------------------------------------
class _SpectralScale(Function):
    """W_sn = W_bar / sigma with sigma = u^T W v treated as a function of W_bar (u, v constants) - discriminator_ap.py:31-32"""

    @staticmethod
    def forward(ctx, w_bar, u, v, sigma, inv_sigma, lazy):
        out = torch.empty_like(w_bar)
        if not lazy:          # lazy: every image the convolutions need comes from the bank's scaled multi-pack; the tensor itself is only a handle
            L.call("hwg_scale_by_ptr", w_bar, inv_sigma, out, w_bar.numel(), _stream())
        ctx.save_for_backward(w_bar, u, v, sigma)
        ctx.leaf = w_bar if isinstance(w_bar, torch.nn.Parameter) and w_bar.requires_grad else None
        return out

    @staticmethod
    def backward(ctx, dwsn):
        w_bar, u, v, sigma = ctx.saved_tensors
        R = w_bar.shape[0]
        K = w_bar.numel() // R
        ws = workspace(L.query("hwg_spectral_workspace", R, K), w_bar.device)
        if ctx.leaf is not None:
            # W_bar is a parameter: its gradient is added where parameter gradients live (the current set or the redirected one), like the
            # convolutions' - returning it would cost the autograd engine an add_ launch per layer and would not follow ops.grad_set
            dst = _grad_buffer(ctx.leaf)
            if DEFER_REDUCE:
                # nothing reads a parameter gradient before the join behind the backward pass: the layers of a network are queued and walk
                # backward together there, two launches instead of two per layer (flush_deferred_reduce -> hwg_spectral_bwd_multi)
                _sn_defer.append((dwsn.contiguous(), w_bar, u, v, sigma, dst, R, K))
                return None, None, None, None, None, None
            L.call("hwg_spectral_bwd", dwsn.contiguous(), w_bar, u, v, sigma, dst, R, K, 1, ws, ws.numel(), _stream())
            return None, None, None, None, None, None
        dwbar = torch.empty_like(w_bar)
        L.call("hwg_spectral_bwd", dwsn.contiguous(), w_bar, u, v, sigma, dwbar, R, K, 0, ws, ws.numel(), _stream())
        return dwbar, None, None, None, None, None


def spectral_normalize(w_bar, u, v, eps=1e-12):
    """one power iteration (u, v updated in place, like the reference does on every forward) and W/sigma"""
    R = w_bar.shape[0]
    K = w_bar.numel() // R
    sig = torch.empty((2,), dtype=torch.float32, device=w_bar.device)
    ws = workspace(L.query("hwg_spectral_workspace", R, K), w_bar.device)
    # the kernels that update u and v in place also write the snapshot this forward's backward pass will read (the reference clones them)
    un, vn = torch.empty_like(u), torch.empty_like(v)
    with torch.no_grad():
        L.call("hwg_spectral_update_to", w_bar.detach(), u, v, un, vn, R, K, eps, sig[0:1], sig[1:2], ws, ws.numel(), _stream())
    w = _SpectralScale.apply(w_bar, un, vn, sig[0:1], sig[1:2], False)
    w._hwg_sn_defer = isinstance(w_bar, torch.nn.Parameter) and w_bar.requires_grad
    return w


class _Prepack:
    """images of one spectral-norm layer's normalised weight for one forward pass (`images`: variant -> tensor), and what to do on a miss"""

    __slots__ = ("images", "bank", "layer", "w_bar", "inv_sigma", "materialised")

    def __init__(self, images, bank, layer, w_bar, inv_sigma, materialised):
        self.images, self.bank, self.layer, self.w_bar, self.inv_sigma, self.materialised = images, bank, layer, w_bar, inv_sigma, materialised

    def miss(self, variant, weight):
        self.bank.request(self.layer, variant)
        if not self.materialised:      # the handle has no values yet: W_bar / sigma into it now, the ordinary pack follows
            with torch.no_grad():
                L.call("hwg_scale_by_ptr", self.w_bar, self.inv_sigma, weight, weight.numel(), _stream())
            self.materialised = True


class SpectralBank:
    """The power iterations of all spectral-norm layers a network runs in one forward pass, as four launches (hwg_spectral_update_multi)
    instead of four per layer. Built once per (network, set of layers): the parameters' storage is persistent, only the snapshot buffer and
    the sigma outputs are fresh per forward."""

    def __init__(self, layers):
        import numpy as np
        self.layers = list(layers)                 # [(w_bar, u, v)] parameters
        rec = np.zeros(len(self.layers), dtype=np.dtype([("W", "<u8"), ("u", "<u8"), ("v", "<u8"), ("copy_off", "<i8"), ("ws_off", "<i8"), ("R", "<i4"), ("K", "<i4")]))
        off = 0
        self.spans = []
        for i, (w, u, v) in enumerate(self.layers):
            R = w.shape[0]
            K = w.numel() // R
            rec[i] = (w.data_ptr(), u.data_ptr(), v.data_ptr(), off, off, R, K)
            self.spans.append((off, R, K))
            off += (R + K + 3) // 4 * 4
        self.total = off
        self.max_R = max(R for _, R, _ in self.spans)
        self.max_K = max(K for _, _, K in self.spans)
        self.ptrs = [(w.data_ptr(), u.data_ptr(), v.data_ptr()) for w, u, v in self.layers]
        dev = self.layers[0][0].device
        self.table = h2d(torch.from_numpy(rec.view(np.uint8)), dev)
        self.ws = torch.empty(self.total, dtype=torch.float32, device=dev)
        # scaled multi-pack: the weight images the layers' convolutions asked for so far, (layer, variant) in request order
        self.requests = []
        self._ptab = None
        self.prepack = bool(int(os.environ.get("HWG_SN_PREPACK", "1") or 0))

    def request(self, layer, variant):
        if (layer, variant) not in self.requests:
            self.requests.append((layer, variant))
            self._ptab = None

    def _pack_table(self):
        import numpy as np
        if self._ptab is None:
            dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("A", "<i4"), ("B", "<i4"), ("Bpad", "<i4"), ("R", "<i4"), ("S", "<i4"), ("flip", "<i4"),
                           ("sa", "<i8"), ("sb", "<i8"), ("sr", "<i8"), ("ss", "<i8"), ("total", "<i8"), ("first_block", "<i8"),
                           ("mode", "<i4"), ("Apad", "<i4"), ("dst_off", "<i8"), ("scale_idx", "<i4"), ("pad", "<i4")])
            assert dt.itemsize == 112
            host = np.zeros(len(self.requests), dtype=dt)
            blocks = off = 0
            spans = []
            for i, (layer, (A, B, Bpad, R, S, sa, sb, flip, wino, Apad)) in enumerate(self.requests):
                total = Apad * Bpad if wino else R * S * A * Bpad
                host[i] = (self.layers[layer][0].data_ptr(), 0, A, B, Bpad, R, S, flip, sa, sb, S, 1, total, blocks, wino, Apad, off, 2 * layer + 1, 0)
                shape = (Bpad // 16, 16, Apad, 16) if wino else (R * S, A, Bpad)
                floats = 16 * Apad * Bpad if wino else total        # (Winograd: one thread per (a, b) pair writes its 16 positions)
                spans.append((off, floats, shape))
                blocks += (total + PACK_PER_BLOCK - 1) // PACK_PER_BLOCK
                off += (floats + 3) // 4 * 4
            self._ptab = (h2d(torch.from_numpy(host.view(np.uint8)), self.ws.device), blocks, off, spans)
        return self._ptab

    def valid(self):
        return all((w.data_ptr(), u.data_ptr(), v.data_ptr()) == p for (w, u, v), p in zip(self.layers, self.ptrs))

    def update(self, eps=1e-12):
        """-> per layer (u snapshot, v snapshot, sigma[0:1], sigma[1:2]) for `spectral_scale`"""
        dev = self.ws.device
        copies = torch.empty(self.total, dtype=torch.float32, device=dev)
        sig = torch.empty(2 * len(self.layers), dtype=torch.float32, device=dev)
        with torch.no_grad():
            L.call("hwg_spectral_update_multi", self.table, len(self.layers), self.max_R, self.max_K, eps, self.ws, copies, sig, _stream())
        images = [{} for _ in self.layers]
        if self.prepack and self.requests:
            tab, blocks, total, spans = self._pack_table()
            buf = torch.empty(total, dtype=torch.float32, device=dev)
            with torch.no_grad():
                L.call("hwg_conv_pack_weight_multi_scaled", tab, len(self.requests), blocks, buf, sig, _stream())
            for (layer, variant), (off, n, shape) in zip(self.requests, spans):
                images[layer][variant] = buf[off: off + n].view(shape)
        out = []
        for i, (off, R, K) in enumerate(self.spans):
            inv = sig[2 * i + 1: 2 * i + 2]
            pre = _Prepack(images[i], self, i, self.layers[i][0], inv, False) if self.prepack else None
            out.append((copies[off: off + R], copies[off + R: off + R + K], sig[2 * i: 2 * i + 1], inv, pre))
        return out


def spectral_scale(w_bar, fresh):
    """W / sigma from a SpectralBank.update() record (the power iteration already ran). With the bank's scaled multi-pack on, the returned
    tensor is a handle: the images the convolutions need were written by that launch and hang on the handle (`_hwg_prepack`); its own
    values are only computed if a convolution asks for an image the bank has not seen yet."""
    un, vn, sigma, inv_sigma, pre = fresh
    lazy = pre is not None and len(pre.images) > 0
    w = _SpectralScale.apply(w_bar, un, vn, sigma, inv_sigma, lazy)
    # read by _Conv2d._wgrad: with deferral on, this weight's gradient is consumed at the flush behind the backward pass (_SpectralScale.backward)
    w._hwg_sn_defer = isinstance(w_bar, torch.nn.Parameter) and w_bar.requires_grad
    if pre is not None:
        pre.materialised = not lazy
        w._hwg_prepack = pre
    return w


# ----------------------------------------------------------------------------------------------
# style extraction helpers
# ----------------------------------------------------------------------------------------------
class _GatherWindows(Function):
    @staticmethod
    def forward(ctx, x, idx_b, idx_pos, window):
        _chk(x, "window source")
        B, Wx, C = x.shape
        n = idx_b.numel()
        out = torch.empty((n, 1, 2 * window + 1, C), dtype=torch.float32, device=x.device)
        L.call("hwg_gather_windows", x, B, Wx, C, idx_b, idx_pos, n, window, out, _stream())
        ctx.save_for_backward(idx_b, idx_pos)
        ctx.cfg = (B, Wx, C, n, window)
        return out

    @staticmethod
    def backward(ctx, dy):
        idx_b, idx_pos = ctx.saved_tensors
        B, Wx, C, n, window = ctx.cfg
        dx = torch.empty((B, Wx, C), dtype=torch.float32, device=dy.device)
        win_of = torch.empty((B * Wx,), dtype=torch.int32, device=dy.device)
        L.call("hwg_scatter_windows", dy.contiguous(), B, Wx, C, idx_b, idx_pos, n, window, win_of, dx, _stream())
        return dx, None, None, None


def gather_windows(x_BWC, idx_b, idx_pos, window):
    """patches[i] = x[idx_b[i], idx_pos[i]-window : idx_pos[i]+window+1, :] (zero outside). PRECONDITION of the deterministic backward
    (hwg_scatter_windows inverts the window list instead of scattering with atomics): the centres (idx_b[i], idx_pos[i]) are unique - true for
    the caller (CharStyleEncoder: one arg-max class per column). HWG_DEBUG=1 checks it on the host."""
    if os.environ.get("HWG_DEBUG"):
        key = (idx_b.long() * x_BWC.shape[1] + idx_pos.long()).cpu()
        assert key.unique().numel() == key.numel(), "gather_windows: duplicate window centres (their gradients would be dropped)"
        assert int(idx_pos.min()) >= 0 and int(idx_pos.max()) < x_BWC.shape[1] and int(idx_b.min()) >= 0 and int(idx_b.max()) < x_BWC.shape[0]
    return _GatherWindows.apply(x_BWC, idx_b, idx_pos, window)


class _SegMean(Function):
    @staticmethod
    def forward(ctx, v, wgt, seg, B):
        n, C = v.shape
        out = torch.empty((B, C), dtype=torch.float32, device=v.device)
        wsum = torch.empty((B,), dtype=torch.float32, device=v.device)
        L.call("hwg_segment_weighted_mean", v, wgt, seg, n, C, B, out, wsum, _stream())
        ctx.save_for_backward(wgt, seg, wsum)
        ctx.cfg = (n, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        wgt, seg, wsum = ctx.saved_tensors
        n, C = ctx.cfg
        dv = torch.empty((n, C), dtype=torch.float32, device=dout.device)
        L.call("hwg_segment_weighted_mean_bwd", dout.contiguous(), wgt, seg, wsum, n, C, dv, _stream())
        return dv, None, None, None


def segment_weighted_mean(v, wgt, seg, B):
    return _SegMean.apply(v.contiguous(), wgt, seg, B)


def gather_scores(x_BWC, idx_b, idx_pos, idx_cls):
    n = idx_b.numel()
    out = torch.empty((n,), dtype=torch.float32, device=x_BWC.device)
    B, Wx, C = x_BWC.shape
    L.call("hwg_gather_scores", x_BWC, B, Wx, C, idx_b, idx_pos, idx_cls, n, out, _stream())
    return out


# ----------------------------------------------------------------------------------------------
# random numbers (device Philox; parity tests inject host-drawn tensors instead)
# ----------------------------------------------------------------------------------------------
class DeviceRNG:
    def __init__(self, seed=0):
        self.seed = int(seed)
        self.offset = 0
        self.text_offset = 0      # `insert_spaces` draws from its own Philox stream (seed + 1): the pipelined generation loop plans request
        # i+1 before it renders request i, and both orders must consume the noise stream and the text stream identically

    def randn(self, shape, device):
        out = torch.empty(shape, dtype=torch.float32, device=device)
        n = out.numel()
        L.call("hwg_randn", out, n, self.seed, self.offset, _stream())
        self.offset += (n + 3) // 4
        return out

    def insert_spaces_begin(self, counts, label, label_lengths, count_std, dup_std, count_duplicates):
        """device `insert_spaces`, first half: counts [L,B,2] float, label [L,B] int32, label_lengths [B] int32 (all on the GPU); draws the plan
        and starts the small device->host copy of the expanded lengths that will size the result"""
        Lc, B = label.shape
        dev = counts.device
        reps = torch.empty((B, 2 * Lc), dtype=torch.int32, device=dev)
        starts = torch.empty((B, Lc), dtype=torch.int32, device=dev)
        lens_max = torch.empty((B + 1,), dtype=torch.int32, device=dev)
        L.call("hwg_insert_spaces_plan", counts.contiguous(), label_lengths, Lc, B, float(count_std), float(dup_std), int(bool(count_duplicates)),
               self.seed + 1, self.text_offset, reps, starts, lens_max, _stream())
        self.text_offset += Lc * B
        return (label, label_lengths, reps, starts, AsyncFetch(lens_max))

    def insert_spaces_finish(self, plan):
        """-> (idx int32 [T,B] on the GPU, padded fractions)"""
        label, label_lengths, reps, starts, fetch = plan
        Lc, B = label.shape
        host = fetch.get().tolist()
        lens, max_count = host[:B], host[B]
        T = max(lens) + max_count
        idx = torch.zeros((T, B), dtype=torch.int32, device=label.device)
        L.call("hwg_insert_spaces_fill", label, label_lengths, reps, starts, Lc, B, T, idx, _stream())
        return idx, [(T - n) / T for n in lens]

    def dropmask(self, shape, p, device):
        out = torch.empty(shape, dtype=torch.float32, device=device)
        n = out.numel()
        L.call("hwg_dropmask", out, n, float(p), self.seed, self.offset, _stream())
        self.offset += (n + 3) // 4
        return out


    def dropmask_multi(self, counts, ps, device):
        """the feature-dropout masks of one network pass from ONE launch: counts[j] elements (multiples of 4) with drop probability ps[j], back
        to back in one buffer -> list of flat views; the values (and the stream position afterwards) are those of consecutive dropmask calls"""
        import numpy as np
        total = int(sum(counts))
        out = torch.empty((total,), dtype=torch.float32, device=device)
        ne = np.asarray(counts, dtype=np.int64); pp = np.asarray(ps, dtype=np.float32)
        L.call("hwg_dropmask_multi", out, len(counts), ne.ctypes.data, pp.ctypes.data, self.seed, self.offset, _stream())
        self.offset += total // 4
        views, off = [], 0
        for n in counts:
            views.append(out[off: off + n]); off += n
        return views


# ----------------------------------------------------------------------------------------------
# frozen BatchNorm (eval) and the FusedUpsample weight transform
# ----------------------------------------------------------------------------------------------
def norm_apply_frozen(x, running_mean, running_var, eps, gamma, beta, act=ACT_NONE, slope=0.0):
    if torch.is_grad_enabled() and (x.requires_grad or (gamma is not None and gamma.requires_grad)):
        raise L.HwgError("eval-mode BatchNorm has no backward kernel; run it under torch.no_grad()")
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    y = torch.empty_like(x)
    mean = torch.empty((N, C), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    L.call("hwg_norm_frozen_fwd", x, y, N, HW, C, running_mean, running_var, eps, gamma, beta, act, slope, mean, rstd, _stream())
    return y


class _FusedUpWeight(Function):
    """[A,B,3,3] -> [A,B,4,4]: the averaged-shift weight of the reference's FusedUpsample (model/pure_gen.py:268-276)"""

    @staticmethod
    def forward(ctx, w3, mult):
        _chk(w3, "fused upsample weight")
        A, B = w3.shape[0], w3.shape[1]
        w4 = torch.empty((A, B, 4, 4), dtype=torch.float32, device=w3.device)
        L.call("hwg_fused_upsample_weight_fwd", w3, w4, A * B, mult, _stream())
        ctx.cfg = (A, B, mult)
        ctx.param_refs = (w3,)
        return w4

    @staticmethod
    def backward(ctx, dw4):
        A, B, mult = ctx.cfg
        (wref,) = ctx.param_refs
        if _direct(wref):      # a parameter: added where parameter gradients live (the current set or the redirected one), no add launch behind it
            L.call("hwg_fused_upsample_weight_bwd_acc", dw4.contiguous(), _grad_buffer(wref), A * B, mult, _stream())
            return None, None
        dw3 = torch.empty((A, B, 3, 3), dtype=torch.float32, device=dw4.device)
        L.call("hwg_fused_upsample_weight_bwd", dw4.contiguous(), dw3, A * B, mult, _stream())
        return dw3, None


def fused_upsample_weight(w3, mult):
    return _FusedUpWeight.apply(w3, float(mult))


class _Scale(Function):
    """y = x * c with a python constant (EqualLR weight scaling, loss weights)"""

    @staticmethod
    def forward(ctx, x, c):
        y = torch.empty_like(x)
        L.call("hwg_axpby", x, float(c), None, 0.0, y, x.numel(), _stream())
        ctx.c = float(c)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = torch.empty_like(dy)
        L.call("hwg_axpby", dy.contiguous(), ctx.c, None, 0.0, dx, dy.numel(), _stream())
        return dx, None


def scale(x, c):
    if float(c) == 1.0:       # (the adversarial losses carry weight 1 in every shipped config: no launch, forward or backward)
        return x
    return _Scale.apply(x.contiguous(), c)


class _WeightedSum(Function):
    """(sum_i w_i x_i, [w_i x_i]) over 0-dim tensors in ONE launch per direction - the trainer's weighted loss accumulation, rounded like the
    chain of scale / add launches it replaces (every product and every partial sum in fp32, left to right; w == 1 multiplies nothing)"""

    @staticmethod
    def forward(ctx, weights, *xs):
        import numpy as np
        n = len(xs)
        xs = [x.contiguous() for x in xs]
        total = torch.empty((), dtype=torch.float32, device=xs[0].device)
        scaled = torch.empty((n,), dtype=torch.float32, device=xs[0].device)
        ptrs = np.array([x.data_ptr() for x in xs], dtype=np.int64)
        w = np.array(weights, dtype=np.float32)
        L.call("hwg_weighted_sum", ptrs.ctypes.data, w.ctypes.data, n, scaled, total, _stream())
        ctx.w = w
        ctx.mark_non_differentiable(scaled)
        return total, scaled

    @staticmethod
    def backward(ctx, g, _unused):
        n = len(ctx.w)
        out = torch.empty((n,), dtype=torch.float32, device=g.device)
        L.call("hwg_weighted_sum_bwd", g.contiguous(), ctx.w.ctypes.data, n, out, _stream())
        return (None,) + tuple(out[i].reshape(()) if ctx.needs_input_grad[1 + i] else None for i in range(n))


def weighted_sum(xs, weights):
    """-> (sum_i weights[i] * xs[i] as a 0-dim tensor, the n scaled terms as a [n] tensor (no gradient))"""
    return _WeightedSum.apply(tuple(float(w) for w in weights), *xs)


class _Add(Function):
    """z = a*x + b*y (residual adds, loss sums)"""

    @staticmethod
    def forward(ctx, x, y, a, b):
        z = torch.empty_like(x)
        L.call("hwg_axpby", x, float(a), y, float(b), z, x.numel(), _stream())
        ctx.ab = (float(a), float(b))
        return z

    @staticmethod
    def backward(ctx, dz):
        a, b = ctx.ab
        dz = dz.contiguous()
        dx = dz if a == 1.0 else scale(dz, a)
        dy = dz if b == 1.0 else scale(dz, b)
        return dx, dy, None, None


def add(x, y, a=1.0, b=1.0):
    assert x.shape == y.shape
    return _Add.apply(x.contiguous(), y.contiguous(), a, b)


class _SplitCols(Function):
    """x [rows, sum(widths)] -> tuple of contiguous [rows, w_i] column blocks"""

    @staticmethod
    def forward(ctx, x, *widths):
        _chk(x, "split input")
        rows, Ct = x.shape
        outs = []
        off = 0
        st = _stream()
        for w in widths:
            o = torch.empty((rows, w), dtype=torch.float32, device=x.device)
            L.call("hwg_copy_channels", x, Ct, off, o, w, 0, w, rows, 1, 0, 0, st)
            outs.append(o)
            off += w
        ctx.cfg = (rows, Ct, widths)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        rows, Ct, widths = ctx.cfg
        dx = torch.empty((rows, Ct), dtype=torch.float32, device=grads[0].device)
        off = 0
        st = _stream()
        for g, w in zip(grads, widths):
            L.call("hwg_copy_channels", g.contiguous(), w, 0, dx, Ct, off, w, rows, 1, 0, 0, st)
            off += w
        return (dx,) + (None,) * len(widths)


def split_cols(x, widths):
    assert sum(widths) == x.shape[1]
    return _SplitCols.apply(x.contiguous(), *widths)


class _ChannelAffine(Function):
    """y = x * scale[c] + shift[c] over the last dim"""

    @staticmethod
    def forward(ctx, x, scale_, shift):
        _chk(x, "channel_affine input"); _chk(scale_, "scale"); _chk(shift, "shift")
        C = x.shape[-1]
        y = torch.empty_like(x)
        L.call("hwg_channel_affine", x, scale_, shift, y, x.numel() // C, C, _stream())
        ctx.save_for_backward(x, scale_)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale_ = ctx.saved_tensors
        dy = dy.contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        st = _stream()
        dx = torch.empty_like(x)
        L.call("hwg_channel_affine", dy, scale_, None, dx, rows, C, st)
        prod = torch.empty_like(x)
        L.call("hwg_mul", dy, x, prod, x.numel(), st)
        ds = colsum(prod.view(rows, C))
        dm = colsum(dy.view(rows, C))
        return dx, ds, dm


def channel_affine(x, scale_, shift):
    return _ChannelAffine.apply(x.contiguous(), scale_.contiguous(), shift.contiguous())


class _Permute(Function):
    """contiguous copy of x.permute(perm) for tensors of rank <= 4 (one strided-gather kernel)"""

    @staticmethod
    def forward(ctx, x, perm):
        _chk(x, "permute input")
        nd = x.dim()
        assert nd <= 4 and sorted(perm) == list(range(nd))
        shape = list(x.shape)
        strides = [1] * nd
        for i in range(nd - 2, -1, -1):
            strides[i] = strides[i + 1] * shape[i + 1]
        odims = [shape[p] for p in perm]
        ostr = [strides[p] for p in perm]
        pad = 4 - nd
        out = permute4(x, [1] * pad + odims, [0] * pad + ostr).view(odims)
        ctx.perm = perm
        return out

    @staticmethod
    def backward(ctx, dy):
        perm = ctx.perm
        inv = [0] * len(perm)
        for i, p in enumerate(perm):
            inv[p] = i
        return _Permute.apply(dy.contiguous(), tuple(inv)), None


def permute(x, perm):
    return _Permute.apply(x.contiguous(), tuple(perm))


def permute_bl_to_lb(x):
    """[B,1,L,C] -> [L,B,C]"""
    B, _, Lr, C = x.shape
    return permute(x.view(B, Lr, C), (1, 0, 2))


def permute_lb_to_bl(x):
    """[L,B,C] -> [B,1,L,C]"""
    Lr, B, C = x.shape
    return permute(x, (1, 0, 2)).view(B, 1, Lr, C)


class _MulConst(Function):
    """y = x * m with m a constant tensor of the same shape (elementwise dropout masks)"""

    @staticmethod
    def forward(ctx, x, m):
        _chk(x, "mul input"); _chk(m, "mul mask")
        y = torch.empty_like(x)
        L.call("hwg_mul", x, m, y, x.numel(), _stream())
        ctx.save_for_backward(m)
        return y

    @staticmethod
    def backward(ctx, dy):
        (m,) = ctx.saved_tensors
        dx = torch.empty_like(m)
        L.call("hwg_mul", dy.contiguous(), m, dx, m.numel(), _stream())
        return dx, None


def mul_const(x, m):
    return _MulConst.apply(x.contiguous(), m)


class _RepeatRows(Function):
    """[n, C] -> [n*k, C], every row repeated k times consecutively (style per author -> per line)"""

    @staticmethod
    def forward(ctx, x, k):
        _chk(x, "repeat_rows input")
        n, C = x.shape
        out = torch.empty((n * k, C), dtype=torch.float32, device=x.device)
        L.call("hwg_copy_channels", x, C, 0, out, C, 0, C, n * k, k, 1, 0, _stream())
        ctx.cfg = (n, C, k)
        return out

    @staticmethod
    def backward(ctx, dy):
        n, C, k = ctx.cfg
        dx = torch.empty((n, C), dtype=torch.float32, device=dy.device)
        L.call("hwg_reduce_rows", dy.contiguous(), C, 0, dx, C, n, k, 0, _stream())
        return dx, None


def repeat_rows(x, k):
    return _RepeatRows.apply(x.contiguous(), int(k))


class _ZeroRowsFrom(Function):
    """y = x with x[pos:] = 0 along dim 0 (the in-place `counts[pos:] = 0` of trainer :697)"""

    @staticmethod
    def forward(ctx, x, pos):
        y = x.clone()
        y[pos:].zero_()
        ctx.pos = pos
        return y

    @staticmethod
    def backward(ctx, dy):
        dx = dy.clone()
        dx[ctx.pos:].zero_()
        return dx, None


def zero_rows_from(x, pos):
    return _ZeroRowsFrom.apply(x, int(pos))


# ----------------------------------------------------------------------------------------------
# bank of linear layers sharing one input (generator AdaIN affines)
# ----------------------------------------------------------------------------------------------
class LinearBank:
    """Evaluates L `Linear(I, O_l)` modules on the same input in one launch (forward) / two launches (backward) through device pointer
    tables; the modules keep their own parameters. Each layer's output comes back split in `halves` contiguous [B, O_l/halves] tensors."""

    def __init__(self, linears, halves=2):
        import numpy as np
        self.linears = list(linears)
        self.halves = halves
        self.L = len(self.linears)
        self.I = self.linears[0].weight.shape[1]
        self.O = [int(m.weight.shape[0]) for m in self.linears]
        assert all(m.weight.shape[1] == self.I for m in self.linears) and all(o % halves == 0 for o in self.O)
        self.first = np.concatenate([[0], np.cumsum(self.O)]).astype(np.int32)
        self.total = int(self.first[-1])
        self._tab = None
        self._key = None
        self._gtab = None
        self._gkey = None

    def tables(self, device, B):
        import numpy as np
        key = (self.linears[0].weight.data_ptr(), str(device), B)
        if self._key != key:
            off = (self.first[:-1].astype(np.int64)) * B
            ints = h2d(np.concatenate([np.array(self.O, dtype=np.int32), self.first]), device)
            ptrs = h2d(np.concatenate([np.array([m.weight.data_ptr() for m in self.linears], dtype=np.int64),
                                       np.array([m.bias.data_ptr() for m in self.linears], dtype=np.int64), off]), device)
            L_ = self.L
            self._tab = (ints[:L_], ints[L_:], ptrs[:L_], ptrs[L_:2 * L_], ptrs[2 * L_:])
            self._key = key
        return self._tab

    def grad_tables(self, device):
        import numpy as np
        # (_grad_buffer also marks the tensors touched for the trainer's None-gradient bookkeeping; frozen parameters get a null entry)
        gw = [_grad_buffer(m.weight) if m.weight.requires_grad else None for m in self.linears]
        gb = [_grad_buffer(m.bias) if m.bias.requires_grad else None for m in self.linears]
        addr = [g.data_ptr() if g is not None else 0 for g in gw + gb]
        key = tuple(addr)
        if self._gkey is None:
            self._gkey = {}
        tab = self._gkey.get(key)           # (one table per destination: the parameters' own gradients, or a stashed set's buffer)
        if tab is None:
            ptrs = h2d(np.array(addr, dtype=np.int64), device)
            tab = self._gkey[key] = (ptrs[:self.L], ptrs[self.L:])
        return tab

    def params(self):
        return [m.weight for m in self.linears] + [m.bias for m in self.linears]

    def __call__(self, x):
        # the parameters are passed to the autograd function although the kernels reach them through pointer tables: autograd must see
        # inputs that require grad, or - when x itself does not (the sampled styles of the text-only "gen" lessons) - it would never call
        # backward and the parameters would silently get no gradient
        outs = _LinearBank.apply(x.contiguous(), self, *self.params())
        h = self.halves
        return [outs[l * h:(l + 1) * h] for l in range(self.L)]


class _LinearBank(Function):
    @staticmethod
    def forward(ctx, x, bank, *params):
        _chk(x, "linear bank input")
        B, I = x.shape
        assert I == bank.I
        O, first, wptr, bptr, off = bank.tables(x.device, B)
        y = torch.empty((bank.total * B,), dtype=torch.float32, device=x.device)
        L.call("hwg_linear_bank_fwd", x, wptr, bptr, O, first, off, bank.L, B, I, bank.halves, bank.total, y, _stream())
        outs = []
        for l in range(bank.L):
            C = bank.O[l] // bank.halves
            base = int(bank.first[l]) * B
            for h in range(bank.halves):
                outs.append(y[base + h * B * C: base + (h + 1) * B * C].view(B, C))
        ctx.save_for_backward(x)
        ctx.bank = bank
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        import numpy as np
        (x,) = ctx.saved_tensors
        bank = ctx.bank
        B, I = x.shape
        O, first, wptr, bptr, off = bank.tables(x.device, B)
        keep = [g.contiguous() if g is not None else None for g in grads]
        dyptr = h2d(np.array([g.data_ptr() if g is not None else 0 for g in keep], dtype=np.int64), x.device)
        gw, gb = bank.grad_tables(x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need = L.query("hwg_linear_bank_bwd_workspace", bank.total, B, I)
        ws = workspace(need, x.device)
        L.call("hwg_linear_bank_bwd", x, dyptr, wptr, gw, gb, O, first, bank.L, B, I, bank.halves, bank.total, dx, ws, ws.numel(), _stream())
        return (dx, None) + (None,) * (2 * bank.L)      # parameter gradients were accumulated in place (grad_tables)


class MLPChain:
    """L square Linear(D, D) modules each followed by LeakyReLU(slope), evaluated in one launch per direction (style embedding MLP)"""

    def __init__(self, linears, slope):
        self.linears = list(linears)
        self.slope = float(slope)
        self.L = len(self.linears)
        self.D = int(self.linears[0].weight.shape[0])
        assert all(tuple(m.weight.shape) == (self.D, self.D) for m in self.linears)
        self._key = self._tab = self._gkey = self._gtab = None

    def tables(self, device):
        import numpy as np
        key = (self.linears[0].weight.data_ptr(), str(device))
        if self._key != key:
            p = h2d(np.array([m.weight.data_ptr() for m in self.linears] + [m.bias.data_ptr() for m in self.linears], dtype=np.int64), device)
            self._tab = (p[:self.L], p[self.L:])
            self._key = key
        return self._tab

    def grad_tables(self, device):
        import numpy as np
        gw = [_grad_buffer(m.weight) if m.weight.requires_grad else None for m in self.linears]
        gb = [_grad_buffer(m.bias) if m.bias.requires_grad else None for m in self.linears]
        addr = [g.data_ptr() if g is not None else 0 for g in gw + gb]
        key = tuple(addr)
        if self._gkey is None:
            self._gkey = {}
        tab = self._gkey.get(key)
        if tab is None:
            p = h2d(np.array(addr, dtype=np.int64), device)
            tab = self._gkey[key] = (p[:self.L], p[self.L:])
        return tab

    def params(self):
        return [m.weight for m in self.linears] + [m.bias for m in self.linears]

    def __call__(self, x):
        return _MLPChain.apply(x.contiguous(), self, *self.params())    # (parameters passed for autograd's sake, see LinearBank.__call__)


MLP_CHAIN_SPLIT = bool(int(os.environ.get("HWG_MLP_SPLIT", "1") or 0))


class _MLPChain(Function):
    @staticmethod
    def forward(ctx, x, chain, *params):
        _chk(x, "mlp chain input")
        B, D = x.shape
        assert D == chain.D
        wptr, bptr = chain.tables(x.device)
        acts = torch.empty((chain.L + 1, B, D), dtype=torch.float32, device=x.device)
        L.call("hwg_mlp_chain_fwd", x, wptr, bptr, chain.L, B, D, chain.slope, acts, _stream())
        ctx.save_for_backward(acts)
        ctx.chain = chain
        return acts[chain.L]

    @staticmethod
    def backward(ctx, dout):
        (acts,) = ctx.saved_tensors
        chain = ctx.chain
        _, B, D = acts.shape
        wptr, _ = chain.tables(acts.device)
        gw, gb = chain.grad_tables(acts.device)
        dx = torch.empty((B, D), dtype=torch.float32, device=acts.device) if ctx.needs_input_grad[0] else None
        if MLP_CHAIN_SPLIT:      # chain of data gradients on one workgroup, then all parameter gradients in parallel (bit-identical, 69 -> ~25 us)
            need = L.query("hwg_mlp_chain_bwd_workspace", chain.L, B, D)
            ws = workspace(need, acts.device)
            L.call("hwg_mlp_chain_bwd_split", dout.contiguous(), acts, wptr, gw, gb, chain.L, B, D, chain.slope, dx, ws, ws.numel(), _stream())
        else:
            L.call("hwg_mlp_chain_bwd", dout.contiguous(), acts, wptr, gw, gb, chain.L, B, D, chain.slope, dx, _stream())
        return (dx, None) + (None,) * (2 * chain.L)
