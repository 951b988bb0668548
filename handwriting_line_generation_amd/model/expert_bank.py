"""All character-style experts evaluated together (reference: model/char_style.py:84-124 run once per class, :210-235).

The 79 `CharExtractor` modules keep their own parameters (state-dict compatible), but the forward/backward for ALL recognised
windows is a fixed sequence of grouped kernels that read each window's expert weights through device pointer tables
(csrc/expert_bank.hip). Parameter gradients are accumulated by the kernels directly into each expert's `.grad` buffer; experts
without a window in the batch are never touched, which preserves the reference's "gradient is None" state for them.
"""
import numpy as np
import torch
from ..ops import Function

from .. import _lib as L
from .. import ops

# (attribute path, is_bias) of every parameter kind of one expert
KINDS = {
    "w1": ("conv1", 1, "weight"), "b1": ("conv1", 1, "bias"), "g1": ("conv1", 2, "weight"), "be1": ("conv1", 2, "bias"),
    "w2": ("conv1", 4, "weight"), "b2": ("conv1", 4, "bias"),
    "w3": ("conv2", 1, "weight"), "b3": ("conv2", 1, "bias"), "g2": ("conv2", 2, "weight"), "be2": ("conv2", 2, "bias"),
    "w4": ("fc", 0, "weight"), "b4": ("fc", 0, "bias"), "w5": ("fc", 2, "weight"), "b5": ("fc", 2, "bias"),
}


def _st():
    return ops._stream()


class ExpertBank:
    """pointer tables over the experts' parameters and gradient buffers"""

    def __init__(self, experts):
        self.experts = experts
        self.E = len(experts)
        self.params = {k: [getattr(getattr(ex, seq)[idx], leaf) for ex in experts] for k, (seq, idx, leaf) in KINDS.items()}
        self._pptr = None
        self._pkey = None
        self._gptr_host = {k: np.zeros(self.E, dtype=np.int64) for k in KINDS}
        self._gptr_dev = {}

    def _static_grads(self, device):
        """True when all expert parameters own a gradient view registered with one FlatParams (then the tables are built once)"""
        st = getattr(self, "_static_state", None)
        if st is None:
            ok, owner, idx = True, None, [[] for _ in range(self.E)]
            for k in KINDS:
                for e, p in enumerate(self.params[k]):
                    fk = getattr(p, "_hwg_flat", None)
                    if fk is None or p.grad is None or (owner is not None and fk[0] is not owner):
                        ok = False
                        break
                    owner = fk[0]
                    idx[e].append(fk[1])
                if not ok:
                    break
            if ok:
                host = np.array([[p.grad.data_ptr() for p in self.params[k]] for k in KINDS], dtype=np.int64)
                self._gptr_static_host = host
                self._set_tabs = {}
                dev = ops.h2d(host, device)
                self._gptr_dev = {k: dev[i] for i, k in enumerate(KINDS)}
                self._flat_idx = (owner, [np.array(v, dtype=np.int64) for v in idx])
                self._grad_key = self.params["w1"][0].grad.data_ptr()
            st = self._static_state = bool(ok)
        if st and self.params["w1"][0].grad is not None and self.params["w1"][0].grad.data_ptr() != self._grad_key:
            self._static_state = None      # gradients were re-created (a new FlatParams / zero_grad(set_to_none)): rebuild
            return self._static_grads(device)
        return st

    def param_ptrs(self, device):
        key = (self.params["w1"][1].data_ptr(), str(device))
        if self._pkey != key:
            host = np.array([[p.data_ptr() for p in self.params[k]] for k in KINDS], dtype=np.int64)
            dev = ops.h2d(host, device)
            self._pptr = {k: dev[i] for i, k in enumerate(KINDS)}
            self._pkey = key
        return self._pptr

    def grad_ptrs(self, plan, device):
        """gradient-buffer tables; allocates / marks-as-touched the buffers of the experts present in this batch.
        The walk over (kinds x present experts) runs once per plan (= once per forward): the touched flags it sets live until the
        trainer's next zero_grad, which is always followed by a new forward and hence a new plan."""
        gs = ops.GRAD_SET
        if gs is None and plan.get("grads_ready") and self._gptr_dev:
            return self._gptr_dev
        present = plan["present"]
        if gs is not None:
            # gradient-set redirect (ops.grad_set): the same offsets of the set's buffer, the set's mask (one table per buffer, built once)
            if not self._static_grads(device):
                raise L.HwgError("expert bank under a gradient-set redirect needs parameters that live in a FlatParams buffer")
            flat, idx = self._flat_idx
            buf, mask = gs
            mask[np.concatenate([idx[e] for e in present])] = True
            tab = self._set_tabs.get(buf.data_ptr())
            if tab is None:
                dev = ops.h2d(self._gptr_static_host + (buf.data_ptr() - flat.flat_grad.data_ptr()), device)
                tab = self._set_tabs[buf.data_ptr()] = {k: dev[i] for i, k in enumerate(KINDS)}
            return tab
        if self._static_grads(device):
            # trainer case: every expert gradient is a persistent view of the flat buffer -> one static pointer table for all experts,
            # and "touched" is one vectorised assignment over the flat indices of the present experts' tensors
            flat, idx = self._flat_idx
            flat.touched[np.concatenate([idx[e] for e in present])] = True
            plan["grads_ready"] = True
            return self._gptr_dev
        changed = False
        for k in KINDS:
            plist = self.params[k]
            hp = self._gptr_host[k]
            for e in present:
                g = ops._grad_buffer(plist[e])
                a = g.data_ptr()
                if hp[e] != a:
                    hp[e] = a
                    changed = True
        if changed or not self._gptr_dev:
            dev = ops.h2d(np.stack([self._gptr_host[k] for k in KINDS]), device)
            self._gptr_dev = {k: dev[i] for i, k in enumerate(KINDS)}
        plan["grads_ready"] = True
        return self._gptr_dev


class _GroupedConv1d(Function):
    @staticmethod
    def forward(ctx, x, bank, wk, bk, plan, S, pad):
        n, _, R, Cin = x.shape
        Cout = bank.params[wk][1].shape[0]
        pp = bank.param_ptrs(x.device)
        y = torch.empty((n, 1, R, Cout), dtype=torch.float32, device=x.device)
        tseg, trow, nt, _ = plan_tiles(plan, R, x.device)
        L.call("hwg_grouped_conv1d_fwd", x, plan["seg_start"], plan["seg_eid"], tseg, trow, nt, pp[wk], pp[bk], y, R, Cin, Cout, S, pad, _st())
        ctx.save_for_backward(x)
        ctx.cfg = (bank, wk, bk, plan, S, pad, n, R, Cin, Cout)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        bank, wk, bk, plan, S, pad, n, R, Cin, Cout = ctx.cfg
        dy = dy.contiguous()
        pp = bank.param_ptrs(x.device)
        gp = bank.grad_ptrs(plan, x.device)
        st = _st()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            tseg, trow, nt, _ = plan_tiles(plan, R, x.device)
            L.call("hwg_grouped_conv1d_dgrad", dy, plan["seg_start"], plan["seg_eid"], tseg, trow, nt, pp[wk], dx, R, Cin, Cout, S, pad, st)
        wseg, wrow, wnt, wrun = plan_tiles(plan, R, x.device, WGRAD_TILE_ROWS)
        need = L.query("hwg_grouped_conv1d_wgrad_workspace", wnt, Cin, Cout, S)
        ws = ops.workspace(need, x.device)
        L.call("hwg_grouped_conv1d_wgrad", dy, x, plan["seg_start"], plan["seg_eid"], plan["G"], wseg, wrow, wrun, wnt, WGRAD_TILE_ROWS, gp[wk], gp[bk], R, Cin, Cout, S, pad,
               ws, ws.numel(), st)
        return dx, None, None, None, None, None, None


class _GroupedGN(Function):
    """GroupNorm + ReLU with each window's affine parameters taken from its expert"""

    @staticmethod
    def forward(ctx, x, bank, gk, bk, plan, groups, eps):
        n, _, R, C = x.shape
        pp = bank.param_ptrs(x.device)
        st = _st()
        gamma = torch.empty((n, C), dtype=torch.float32, device=x.device)
        beta = torch.empty((n, C), dtype=torch.float32, device=x.device)
        L.call("hwg_gather_rows_ptr", pp[gk], plan["eid"], gamma, n, C, st)
        L.call("hwg_gather_rows_ptr", pp[bk], plan["eid"], beta, n, C, st)
        y = torch.empty_like(x)
        mean = torch.empty((n, C), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        ws = ops.workspace(L.query("hwg_norm_workspace", n, R, C), x.device)
        L.call("hwg_norm_fwd", x, y, n, R, C, ops.NORM_GN, groups, eps, gamma, beta, 1, None, ops.ACT_RELU, 0.0, mean, rstd, None, None, 0.0,
               ws, ws.numel(), st)
        ctx.save_for_backward(x, y, gamma, beta, mean, rstd)
        ctx.cfg = (bank, gk, bk, plan, groups, n, R, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, beta, mean, rstd = ctx.saved_tensors
        bank, gk, bk, plan, groups, n, R, C = ctx.cfg
        dy = dy.contiguous()
        st = _st()
        dx = torch.empty_like(x)
        dgamma = torch.empty((n, C), dtype=torch.float32, device=x.device)
        dbeta = torch.empty_like(dgamma)
        ws = ops.workspace(L.query("hwg_norm_workspace", n, R, C), x.device)
        L.call("hwg_norm_bwd", dy, x, y, dx, n, R, C, ops.NORM_GN, groups, gamma, beta, 1, None, ops.ACT_RELU, 0.0, mean, rstd, dgamma, dbeta, 0,
               ws, ws.numel(), st)
        gp = bank.grad_ptrs(plan, x.device)
        L.call("hwg_segment_accumulate_ptr", dgamma, plan["seg_start"], plan["seg_eid"], plan["G"], gp[gk], C, st)
        L.call("hwg_segment_accumulate_ptr", dbeta, plan["seg_start"], plan["seg_eid"], plan["G"], gp[bk], C, st)
        return dx, None, None, None, None, None, None


TILE_ROWS = 32          # GT_ROWS of csrc/expert_bank.hip (forward / data gradient)
WGRAD_TILE_ROWS = 64    # rows per weight-gradient work tile (tools/expert_probe.py: 64 beats 128 / 256 - the row loop is a latency chain)


def _tiles_host(starts, R, tile_rows):
    rows = np.diff(starts).astype(np.int64) * R
    per_run = (rows + tile_rows - 1) // tile_rows
    tile_seg = np.repeat(np.arange(rows.size, dtype=np.int32), per_run)
    first = np.concatenate([[0], np.cumsum(per_run)])
    tile_row0 = ((np.arange(tile_seg.size) - np.repeat(first[:-1], per_run)) * tile_rows).astype(np.int32)
    return tile_seg, tile_row0, first.astype(np.int32)


def make_plan(cls_sorted_np, device, window_rows=(), extra=()):
    """cls_sorted_np: int array of the expert id of every window, sorted ascending. `window_rows`: the R values (positions per window)
    the plan will be used with - their work-tile lists (forward/dgrad and weight-gradient tilings) are built here and travel to the
    device in the SAME single upload as the run tables and the caller's `extra` int32 arrays (returned as plan["extra"])."""
    n = cls_sorted_np.size
    change = np.nonzero(np.diff(cls_sorted_np))[0] + 1
    starts = np.concatenate([[0], change, [n]]).astype(np.int32)
    seg_eid = cls_sorted_np[starts[:-1]].astype(np.int32)
    pieces = [np.ascontiguousarray(a, dtype=np.int32).ravel() for a in extra] + [cls_sorted_np.astype(np.int32), starts, seg_eid]
    tile_keys = []
    for R in window_rows:
        for tr in (TILE_ROWS, WGRAD_TILE_ROWS):
            tile_keys.append((R, tr))
            pieces.extend(_tiles_host(starts, R, tr))
    packed = ops.h2d(np.concatenate(pieces), device)
    cuts = np.cumsum([0] + [p.size for p in pieces])
    part = [packed[cuts[i]:cuts[i + 1]] for i in range(len(pieces))]
    ne = len(extra)
    plan = {"extra": part[:ne], "eid": part[ne], "seg_start": part[ne + 1], "seg_eid": part[ne + 2], "G": int(seg_eid.size),
            "present": [int(e) for e in seg_eid], "n": n, "starts_host": starts, "tiles": {}}
    for j, key in enumerate(tile_keys):
        tseg, trow, first = part[ne + 3 + 3 * j: ne + 6 + 3 * j]
        plan["tiles"][key] = (tseg, trow, int(tseg.numel()), first)
    return plan


def plan_tiles(plan, R, device, tile_rows=TILE_ROWS):
    """work list of the grouped GEMMs for windows of R positions: (run, first row) of every tile of at most `tile_rows` rows, and the
    first tile of every run (built by make_plan for the announced R values, lazily otherwise)"""
    key = (R, tile_rows)
    hit = plan["tiles"].get(key)
    if hit is None:
        tile_seg, tile_row0, first = _tiles_host(plan["starts_host"], R, tile_rows)
        nt = int(tile_seg.size)
        packed = ops.h2d(np.concatenate([tile_seg, tile_row0, first]), device)
        hit = plan["tiles"][key] = (packed[:nt], packed[nt:2 * nt], nt, packed[2 * nt:])
    return hit


def run_experts(bank, patches, plan, groups1, groups2, eps=1e-5):
    """patches [n,1,R,C] -> per-window styles [n, style_dim]; the network of char_style.py:118-124 (window < 3 variant)"""
    n, _, R, _ = patches.shape
    h = ops.relu(patches)
    h = _GroupedConv1d.apply(h, bank, "w1", "b1", plan, 3, 1)
    h = _GroupedGN.apply(h, bank, "g1", "be1", plan, groups1, eps)
    h = _GroupedConv1d.apply(h, bank, "w2", "b2", plan, 3, 1)
    h = ops.relu(ops.add(h, patches))
    h = _GroupedConv1d.apply(h, bank, "w3", "b3", plan, 1, 0)
    h = _GroupedGN.apply(h, bank, "g2", "be2", plan, groups2, eps)
    h = ops.avg_pool2d(h, (1, R))                                   # [n,1,1,C]
    h = ops.relu(_GroupedConv1d.apply(h, bank, "w4", "b4", plan, 1, 0))
    h = _GroupedConv1d.apply(h, bank, "w5", "b5", plan, 1, 0)
    return h.reshape(n, -1)
