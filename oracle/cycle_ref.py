"""ORACLE (test infrastructure only): CPU rendition of one 7-lesson curriculum cycle, used ONLY as the timed
`cpu_baseline` of bench.py ("kind": "port") and never by the product path.

It strings the oracle networks (oracle/torch_ref.py) together in the order the reference's trainer runs them
(trainer/hw_with_style_trainer.py:207-418, 514-892): count / gen(no-step) / auto+auto-gen / disc, with the three separate
backward passes of the balanced 'auto' lesson, gradient clipping and Adam steps. The per-parameter Python loops of the
reference's gradient balancing are not reproduced (they add host time, not arithmetic), so this baseline is, if anything,
faster than the reference itself on the same cores.
"""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import seq_oracle, torch_ref


def _leaf(sd, prefix, trainable):
    """copy of the sub-network's state-dict entries; entries named in `trainable` become autograd leaves"""
    out = {}
    for k, v in sd.items():
        if k.startswith(prefix):
            t = v.detach().clone()
            if k in trainable and t.dtype.is_floating_point:
                t.requires_grad_(True)
            out[k[len(prefix):]] = t
    return out


class CycleRef:
    def __init__(self, model_sd, trainable, encoder_sd, B, A, W, label_len, num_class=80, style_dim=128, seed=0):
        self.B, self.A, self.W, self.L, self.C, self.S = B, A, W, label_len, num_class, style_dim
        self.g = torch.Generator().manual_seed(seed)
        self.sd = {name: _leaf(model_sd, name + ".", trainable) for name in ("generator", "discriminator", "hwr", "style_extractor", "spacer")}
        self.enc = {k: v.detach().clone() for k, v in encoder_sd.items()}
        main = [t for n in ("generator", "style_extractor", "spacer") for t in self.sd[n].values() if t.requires_grad]
        disc = [t for t in self.sd["discriminator"].values() if t.requires_grad]
        self.opt = torch.optim.Adam(main, lr=2e-4, betas=(0.5, 0.999))
        self.opt_d = torch.optim.Adam(disc, lr=2e-4, betas=(0.5, 0.999))
        self.all = main + disc + [t for t in self.sd["hwr"].values() if t.requires_grad]
        self.prev_style = None

    def _batch(self):
        img = torch.rand(self.B, 1, 64, self.W, generator=self.g) * 2 - 1
        lab = torch.randint(1, self.C, (self.L, self.B), generator=self.g)
        return img, lab

    def _style(self, img):
        pred = torch_ref.hwr(self.sd["hwr"], img)
        na = self.B // self.A
        T = pred.shape[0]
        cimg = img.view(na, self.A, 64, self.W).permute(0, 2, 1, 3).reshape(na, 1, 64, self.A * self.W)
        crec = pred.permute(1, 0, 2).reshape(na, self.A * T, self.C).permute(0, 2, 1)
        st = torch_ref.style_extractor(self.sd["style_extractor"], cimg, crec, n_class=self.C)
        return pred, st.repeat_interleave(self.A, dim=0)

    def _gen_from_text(self, lab, style):
        oh = F.one_hot(lab, self.C).float()
        counts = torch_ref.spacer(self.sd["spacer"], oh, style)
        spaced, _ = seq_oracle.insert_spaces(lab, [self.L] * self.B, counts, self.C, 1e-8, 1e-9)
        return torch_ref.generator(self.sd["generator"], spaced, style)

    def _ctc(self, pred, lab):
        T = pred.shape[0]
        l = F.ctc_loss(pred, lab.t(), torch.full((self.B,), T, dtype=torch.long), torch.full((self.B,), self.L, dtype=torch.long))
        return torch.where(torch.isinf(l), torch.zeros_like(l), l)

    def _zero(self):
        for t in self.all:
            t.grad = None

    def _clip(self):
        torch.nn.utils.clip_grad_value_(self.all, 2)

    def lesson_count(self):
        self._zero()
        img, lab = self._batch()
        pred, style = self._style(img)
        aligned = seq_oracle.correct_pred_c(pred, lab)
        gt, pos = seq_oracle.gt_counts_c(aligned, lab)
        counts = torch_ref.spacer(self.sd["spacer"], F.one_hot(lab, self.C).float(), style)
        counts = torch.cat((counts[:pos], torch.zeros_like(counts[pos:])), 0)
        (0.5 * F.mse_loss(counts, gt)).backward()
        self._clip(); self.opt.step()

    def lesson_gen(self):
        _, lab = self._batch()
        style = torch.randn(self.B, self.S, generator=self.g) if self.prev_style is None else self.prev_style
        gen = self._gen_from_text(lab, style)
        (1e-4 * self._ctc(torch_ref.hwr(self.sd["hwr"], gen), lab)).backward(retain_graph=True)
        outs = torch_ref.discriminator(self.sd["discriminator"], gen)
        (-(outs[0].mean() + outs[1].mean()) / 2).backward()

    def lesson_auto(self):
        self._zero()
        img, lab = self._batch()
        pred, style = self._style(img)
        self.prev_style = style.detach()
        aligned = seq_oracle.correct_pred_c(pred, lab)
        recon = torch_ref.generator(self.sd["generator"], F.one_hot(aligned, self.C).float(), style)
        Wr = recon.shape[3]
        rp = F.pad(recon, (0, max(self.W - Wr, 0)), value=-1.0)[..., : max(self.W, Wr)]
        ip = F.pad(img, (0, max(Wr - self.W, 0)), value=-1.0)
        outs = torch_ref.discriminator(self.sd["discriminator"], recon)
        (-(outs[0].mean() + outs[1].mean()) / 2).backward(retain_graph=True)
        (1e-6 * self._ctc(torch_ref.hwr(self.sd["hwr"], recon), lab)).backward(retain_graph=True)
        d = ip.shape[3] - recon.shape[3]
        rz = F.pad(recon, (d // 2, d // 2 + d % 2)) if d > 0 else recon
        code, mid = torch_ref.encoder2(self.enc, torch.cat((ip, rz), 0))
        perc = F.l1_loss(code[self.B:], code[:self.B]) + F.l1_loss(mid[self.B:], mid[:self.B])
        (0.5 * F.l1_loss(rp, ip) + 0.5 * perc).backward()
        self._clip(); self.opt.step()

    def lesson_disc(self):
        self._zero()
        img, lab = self._batch()
        style = torch.randn(self.B, self.S, generator=self.g) if self.prev_style is None else self.prev_style
        with torch.no_grad():
            fake = self._gen_from_text(lab, style)
        Wf = fake.shape[3]
        if Wf > self.W:
            img = F.pad(img, (0, Wf - self.W, 0, 0), mode="replicate")
        elif Wf < self.W:
            fake = F.pad(fake, (0, self.W - Wf, 0, 0), mode="replicate")
        outs = torch_ref.discriminator(self.sd["discriminator"], torch.cat((img, fake), 0))
        loss = sum(F.relu(1.0 - o[:self.B]).mean() + F.relu(1.0 + o[self.B:]).mean() for o in outs) / len(outs)
        loss.backward()
        self._clip(); self.opt_d.step()

    def cycle(self):
        """count, gen, auto, disc, gen, auto, disc  (configs/cf_IAMslant_..._sMG.json 'curriculum')"""
        for fn in (self.lesson_count, self.lesson_gen, self.lesson_auto, self.lesson_disc, self.lesson_gen, self.lesson_auto, self.lesson_disc):
            fn()


def time_cycles(model_sd, trainable, encoder_sd, B, A, W, label_len, budget_s=25.0, max_cycles=3):
    """-> (steps_per_sec, n_steps, seconds): whole 7-step cycles until `budget_s` is used (at least one)"""
    ref = CycleRef(model_sd, trainable, encoder_sd, B, A, W, label_len)
    t0 = time.perf_counter()
    n = 0
    while n < max_cycles:
        ref.cycle()
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return 7 * n / dt, 7 * n, dt


def effective_cores():
    """CPUs this process may really use: min(affinity mask, cgroup CPU quota)"""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


if __name__ == "__main__":
    # CLI used by bench.py's cpu_baseline leg (child process, never touches the GPU):
    #   python -m oracle.cycle_ref B A W label_len budget_seconds
    import json
    import sys
    import warnings
    warnings.filterwarnings("ignore")
    B, A, W, Lr = (int(v) for v in sys.argv[1:5])
    budget = float(sys.argv[5])
    cores = effective_cores()
    torch.set_num_threads(cores)
    from handwriting_line_generation_amd.model import Autoencoder, HWWithStyle   # parameter shapes only; nothing is executed through it
    import os
    cfg = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "model_config_iam.json")))
    torch.manual_seed(0)
    m = HWWithStyle(cfg)
    ae = Autoencoder({"type": "2tight", "hwr": cfg["num_class"]})
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    enc = {k[8:]: v for k, v in ae.state_dict().items() if k.startswith("encoder.")}
    sps, nsteps, secs = time_cycles(m.state_dict(), trainable, enc, B, A, W, Lr, budget_s=budget, max_cycles=2)
    print(json.dumps({"steps_per_sec": sps, "steps": nsteps, "seconds": secs, "cores": cores, "torch": torch.__version__}))
