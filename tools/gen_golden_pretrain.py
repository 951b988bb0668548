"""Goldens recorded from the UNMODIFIED reference for the two pre-training trainers and for validation / CER:

  pretrain_auto.json   trainer/auto_trainer.py AutoTrainer (cf_IAM_auto_2tight_newCTC: Autoencoder 2tight, L1 + CTC, clip 2, Adam):
                       2 training iterations (:79-177, run_gen :255-319) + one _valid_epoch over 3 batches (:199-245: val_*, CER, WER)
  pretrain_hwr.json    trainer/hw_with_style_trainer.py HWWithStyleTrainer without curriculum (cf_IAM_hwr_cnnOnly_batchnorm_aug):
                       2 training iterations through run_hwr (:494-512), logged CER / WER included. (Its validation path cannot be
                       recorded: _valid_epoch unpacks run_hwr's 2-tuple into 3 names and raises, :461.)
  valid_gan.json       HWWithStyleTrainer._valid_epoch (:437-486) of the GAN config over 3 batches (curriculum.getValid() lesson),
                       and getCER (:894-914, utils/error_rates.py:2-26) known-answer cases

Per training iteration: logged losses, fingerprints ([sum, sum|.|, sum.^2, cosine projection]) of every parameter's gradient at the point
the trainer clips it (AutoTrainer) or hands it to Adam (recogniser pre-training) and of every parameter's update - from the native fp32 run and from the
run widened to fp64 with identical draws (the yardstick, as in tools/gen_golden_lessons.py).

editdistance (absent here) is replaced by a plain Levenshtein distance - the only thing the reference uses it for.

Build container only:   python tools/gen_golden_pretrain.py [auto hwr valid_gan]
"""
import json
import os
import random
import subprocess
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(ROOT, "tests", "golden")
SEEDS = {"auto": 42, "hwr": 41, "valid_gan": 21, "enc": 22}
SHAPES = {"auto": dict(B=3, W=132, L=5), "hwr": dict(B=4, W=128, L=5), "valid_gan": dict(B=2, A=2, W=256, L=12)}
VALID_BATCHES = (50, 51, 52)


def levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def boot(wide):
    import ref_bootstrap
    ref_bootstrap.bootstrap()
    sys.modules["editdistance"].eval = levenshtein
    import torch
    torch.set_num_threads(8)
    cast = (lambda t: t)
    if wide:
        torch.set_default_dtype(torch.float64)
        cast = (lambda t: t.double())

        class _Wide(torch.Tensor):
            def normal_(self, *a, **k):
                r = torch.empty(self.shape, dtype=torch.float32).normal_(*a, **k)
                return self.as_subclass(torch.Tensor).copy_(r)

        torch.FloatTensor = lambda *a: torch.DoubleTensor(*a).as_subclass(_Wide)
        _rl = torch.randn_like
        torch.randn_like = lambda t, **k: _rl(t.to(torch.float32), **k).double()
        torch.Tensor.float = lambda self, *a, **k: self.double()
    return cast


def seeded(model, seed, wide):
    import torch
    from oracle import torch_ref
    dt = torch.get_default_dtype()
    torch.set_default_dtype(torch.float32)
    sd = torch_ref.seeded_state_dict(model, seed)
    torch.set_default_dtype(dt)
    return {k: (v.double() if wide and v.dtype.is_floating_point else v) for k, v in sd.items()}


def train_iterations(trainer, model, sd0, n_iter, rms_in):
    """n_iter reference iterations, each from the seeded weights `sd0` and an empty optimizer state (teacher forcing, as
    tools/gen_golden_tf.py): recorded are the gradients where the trainer clips them / hands them to Adam, and the parameter updates of an
    Adam step whose moments are seeded draws scaled by each tensor's gradient RMS (oracle/tf_state.py) - the smooth mid-run update
    instead of the sign-like first step. `rms_in`: the RMS values of the fp32 run when this is the widened run."""
    import torch
    from gen_golden_lessons import fingerprint
    from oracle import tf_state
    plist = [p for _, p in model.named_parameters()]
    grads, rms_out = {}, {}
    cur = [0]
    orig_step = trainer.optimizer.step

    def record():
        it = cur[0]
        if it in grads:
            return
        grads[it] = [fingerprint(p.grad, i) if p.grad is not None else None for i, p in enumerate(plist)]
        rms = rms_in[it] if rms_in is not None else [float(p.grad.to(torch.float32).pow(2).mean().sqrt()) if p.grad is not None else None for p in plist]
        rms_out[it] = rms
        for i, p in enumerate(plist):
            if p.grad is not None:
                m, v = tf_state.seeded_moments(p.shape, rms[i], tf_state.moment_key(it, i))
                trainer.optimizer.state[p] = {"step": torch.tensor(float(tf_state.ADAM_STEP)), "exp_avg": m.to(p.dtype), "exp_avg_sq": v.to(p.dtype)}

    def spy_step(*a, **k):          # recogniser pre-training hands the raw gradients to Adam (trainer :389-391) ...
        record()
        return orig_step(*a, **k)
    trainer.optimizer.step = spy_step
    import torch.nn.utils as nnu
    orig_clip = nnu.clip_grad_value_

    def spy_clip(params, value):    # ... the autoencoder trainer clips them first (auto_trainer.py:135): recorded BEFORE the clip, where the
        record()                    # HIP trainer's pre_clip_hook reads them
        return orig_clip(params, value)
    nnu.clip_grad_value_ = spy_clip
    torch.nn.utils.clip_grad_value_ = spy_clip
    its = []
    for it in range(n_iter):
        cur[0] = it
        model.load_state_dict(sd0)
        trainer.optimizer.state.clear()
        for p in plist:
            p.grad = None
        torch.manual_seed(7 + it); np.random.seed(7 + it); random.seed(7 + it)
        snap = [p.detach().clone() for p in plist]
        log = trainer._train_iteration(it)
        upd = [fingerprint(p.detach() - s, i) for i, (p, s) in enumerate(zip(plist, snap))]
        its.append({"log": {k: float(v) for k, v in log.items() if isinstance(v, (int, float)) or hasattr(v, "item")},
                    "grads": grads[it], "rms": rms_out[it], "update": [u if u[1] != 0.0 else None for u in upd]})
        print(it, its[-1]["log"], flush=True)
    return its


def run(case, wide, out_path):
    warnings.filterwarnings("ignore")
    rms_path = "/tmp/hwg_pretrain_%s_rms.json" % case
    cast = boot(wide)
    import torch
    from gen_golden_lessons import _Loader
    from handwriting_line_generation_amd.data.synthetic import SyntheticAuthorDataset
    from handwriting_line_generation_amd.harness import CHAR_FILES, load_config, synthetic_gan_config
    from model import Autoencoder, HWWithStyle
    import model.loss as ref_loss
    from trainer import AutoTrainer, HWWithStyleTrainer
    sh = SHAPES[case]
    out = {"case": case, "seed": SEEDS[case], **sh}
    work = "/tmp/hwg_golden_pretrain_%s_%d" % (case, int(wide))
    os.makedirs(work, exist_ok=True)
    if case in ("auto", "hwr"):
        cfg = load_config("iam_auto" if case == "auto" else "iam_hwr")
        cfg["cuda"] = False
        cfg["data_loader"]["char_file"] = CHAR_FILES["iam"]
        cfg["data_loader"]["batch_size"] = sh["B"]
        cfg["trainer"]["save_dir"] = os.path.join(work, "saved")
        model = (Autoencoder if case == "auto" else HWWithStyle)(cfg["model"])
        sd0 = seeded(model, SEEDS[case], wide)
        model.load_state_dict(sd0)
        if wide:
            model = model.double()
        ds = SyntheticAuthorDataset(CHAR_FILES["iam"], sh["B"], 1, width=sh["W"], label_len=sh["L"])
        losses = {n: getattr(ref_loss, f) for n, f in cfg["loss"].items()}
        cls = AutoTrainer if case == "auto" else HWWithStyleTrainer
        trainer = cls(model, losses, [], None, cfg, _Loader(ds, cast), None, None)
        trainer.logged = True
        out["names"] = [k for k, _ in model.named_parameters()]
        rms_in = {int(k): v for k, v in json.load(open(rms_path)).items()} if wide else None
        out["iterations"] = train_iterations(trainer, model, sd0, 2, rms_in)
        if not wide:
            with open(rms_path, "w") as f:
                json.dump({it: x["rms"] for it, x in enumerate(out["iterations"])}, f)
        model.load_state_dict(sd0)
        if case == "auto":
            batches = []
            for i in VALID_BATCHES:
                b = ds.batch(i)
                b["image"] = cast(b["image"])
                batches.append(b)
            trainer.valid_data_loader = batches
            torch.manual_seed(77); np.random.seed(77); random.seed(77)
            out["valid"] = {k: float(v) for k, v in trainer._valid_epoch().items()}
            print("valid", out["valid"], flush=True)
    else:
        cfg, _ = synthetic_gan_config("iam_gan", sh["B"], sh["A"], workdir=work)
        cfg["cuda"] = False
        ae = Autoencoder({"type": "2tight", "hwr": cfg["model"]["num_class"]})
        torch.save({"state_dict": seeded(ae, SEEDS["enc"], False)}, cfg["trainer"]["encoder_weights"])
        model = HWWithStyle(cfg["model"])
        model.load_state_dict(seeded(model, SEEDS[case], wide))
        if wide:
            model = model.double()
        ds = SyntheticAuthorDataset(cfg["data_loader"]["char_file"], sh["B"], sh["A"], width=sh["W"], label_len=sh["L"])
        losses = {n: getattr(ref_loss, f) for n, f in cfg["loss"].items()}
        trainer = HWWithStyleTrainer(model, losses, [], None, cfg, _Loader(ds, cast), None, None)
        if wide:
            trainer.encoder = trainer.encoder.double()
        trainer.logged = True
        batches = []
        for i in VALID_BATCHES:
            b = ds.batch(i)
            b["image"] = cast(b["image"])
            batches.append(b)
        trainer.valid_data_loader = batches
        out["valid_lesson"] = sorted(trainer.curriculum.getValid())
        torch.manual_seed(77); np.random.seed(77); random.seed(77)
        out["valid"] = {k: float(v) for k, v in trainer._valid_epoch().items()}
        print("valid", out["valid"], flush=True)
        if not wide:
            # getCER known answers: greedy CTC decode of [T, B, C] scores (collapse repeats, drop blanks) against ground-truth strings
            from oracle import cer_kats
            kats = []
            for pred, texts, casesens in cer_kats.cases(trainer.idx_to_char, trainer.num_class):
                trainer.casesensitive = casesens
                cer, wer, strs = trainer.getCER(texts, pred)
                kats.append({"casesensitive": casesens, "cer": float(cer), "wer": float(wer), "strs": strs})
            out["cer_kats"] = kats
    with open(out_path, "w") as f:
        json.dump(out, f)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--child":
        return run(sys.argv[2], sys.argv[3] == "1", sys.argv[4])
    for case in (sys.argv[1:] or ["auto", "hwr", "valid_gan"]):
        res = {}
        for wide in (0, 1):
            tmp = "/tmp/hwg_pretrain_%s_%d.json" % (case, wide)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", case, str(wide), tmp])
            res[wide] = json.load(open(tmp))
        out = dict(res[0])
        out["valid64"] = res[1].get("valid")
        if "iterations" in out:
            for a, b in zip(out["iterations"], res[1]["iterations"]):
                a["log64"], a["grads64"], a["update64"] = b["log"], b["grads"], b["update"]
        name = "valid_gan.json" if case == "valid_gan" else "pretrain_%s.json" % case
        with open(os.path.join(GOLD, name), "w") as f:
            json.dump(out, f, separators=(",", ":"))
        print("wrote", name, os.path.getsize(os.path.join(GOLD, name)) // 1024, "KiB")


if __name__ == "__main__":
    main()
