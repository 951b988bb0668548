"""GPU: the pipelined generation loop (generate.generate_stream) produces exactly what calling the model request by request does."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stream_equals_sequential(cuda, tmp_path):
    from handwriting_line_generation_amd import ops, rng
    from handwriting_line_generation_amd.generate import generate_stream
    from handwriting_line_generation_amd.harness import build_gan_trainer
    torch.manual_seed(0)
    trainer, cfg = build_gan_trainer("iam_gan", 1, 2, width=256, label_len=12, workdir=str(tmp_path))
    model = trainer.model
    model.eval()
    g = torch.Generator().manual_seed(3)
    B = 5
    reqs = []
    for L in (9, 14, 11):
        label = torch.randint(1, cfg["model"]["num_class"], (L, B), generator=g, dtype=torch.int32)
        lens = torch.IntTensor([L, L - 2, L, 3, L - 1])
        style = ops.h2d(torch.randn(B, cfg["model"]["style_dim"], generator=g), trainer.gpu)
        reqs.append((label, lens, style))
    with torch.no_grad():
        rng.set_mode("device", seed=11); np.random.seed(5)
        seq = [model(ops.h2d(l, trainer.gpu), n, s).cpu() for l, n, s in reqs]
        rng.set_mode("device", seed=11); np.random.seed(5)
        stream = [img.cpu() for img, _ in generate_stream(model, reqs)]
    assert len(seq) == len(stream) == 3
    for a, b in zip(seq, stream):
        assert a.shape == b.shape and torch.equal(a, b)


def test_device_insert_spaces(cuda, tmp_path):
    """`insert_spaces` with the device generator: (1) with zero noise it is the host function exactly (same rounding of the same counts);
    (2) with noise every line still spells its text, run lengths are non-negative, the tail padding rule holds, and the draws follow
    N(count, std) on average"""
    from handwriting_line_generation_amd import ops, rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    torch.manual_seed(0)
    trainer, cfg = build_gan_trainer("iam_gan", 1, 2, width=256, label_len=12, workdir=str(tmp_path))
    model = trainer.model
    g = torch.Generator().manual_seed(7)
    L, B = 17, 6
    label = torch.randint(1, cfg["model"]["num_class"], (L, B), generator=g, dtype=torch.int32)
    lens = [17, 12, 1, 9, 17, 5]
    counts = (torch.rand(L, B, 2, generator=g) * 3.2).to(trainer.gpu)
    std = (model.count_std, model.dup_std)
    rng.set_mode("device", seed=5)
    try:
        model.count_std = model.dup_std = 0.0
        idx_h, pad_h = model.insert_spaces_index(label, lens, counts)
        idx_d, pad_d = model.insert_spaces_device(label, lens, counts)
        assert idx_d.dtype == torch.int32 and idx_d.is_cuda
        assert np.array_equal(idx_d.cpu().numpy().astype(np.int64), idx_h) and pad_d == pad_h
        model.count_std, model.dup_std = 0.7, 0.4
        tot_blank, tot_dup, n = 0.0, 0.0, 0
        for rep in range(40):
            idx, pad = model.insert_spaces_device(label, lens, counts)
            a = idx.cpu().numpy()
            T = a.shape[0]
            for b in range(B):
                col = a[:, b]
                body = T - int(round(pad[b] * T))
                assert (col[body:] == 0).all()
                # collapse repeats and drop blanks -> the text (a repeat count of 0 drops a character, so compare as a subsequence)
                runs = [int(col[t]) for t in range(body) if col[t] != 0 and (t == 0 or col[t] != col[t - 1])]
                text = [int(c) for c in label[:lens[b], b]]
                it = iter(text)
                assert all(any(c == d for d in it) for c in runs), (runs, text)
            tot_blank += float((a == 0).sum()) - sum(round(p * T) for p in pad)
            tot_dup += float((a != 0).sum())
            n += 1
        exp_blank = float(sum(counts[:lens[b], b, 0].clamp_min(0).sum() for b in range(B)))
        exp_dup = float(sum(counts[:lens[b], b, 1].clamp_min(0).sum() for b in range(B)))
        assert abs(tot_blank / n - exp_blank) < 0.15 * exp_blank and abs(tot_dup / n - exp_dup) < 0.15 * exp_dup
        # a different seed gives a different expansion, the same seed the same one
        rng.set_mode("device", seed=6); a1 = model.insert_spaces_device(label, lens, counts)[0].cpu()
        rng.set_mode("device", seed=6); a2 = model.insert_spaces_device(label, lens, counts)[0].cpu()
        rng.set_mode("device", seed=9); a3 = model.insert_spaces_device(label, lens, counts)[0].cpu()
        assert torch.equal(a1, a2) and (a1.shape != a3.shape or not torch.equal(a1, a3))
    finally:
        model.count_std, model.dup_std = std
        rng.set_mode("device")


def _oracle_style(sd, image, label, a_batch_size, use_pred, n_class=80):
    """recogniser -> (log-probs | one-hot DTW alignment) -> lines of an author side by side -> style extractor (generate.py:58-81)"""
    import torch.nn.functional as F
    from oracle import seq_oracle, torch_ref
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}   # noqa: E731
    pred = torch_ref.hwr(sub("hwr."), image)
    if use_pred:
        spaced = pred.permute(1, 2, 0)
    else:
        spaced = F.one_hot(seq_oracle.correct_pred(pred, label), n_class).float().permute(1, 2, 0)
    B, feats, h, w = image.shape
    T = spaced.shape[2]
    ci = image.permute(1, 2, 0, 3).contiguous().view(feats, h, B // a_batch_size, w * a_batch_size).permute(2, 0, 1, 3)
    cl = spaced.permute(1, 0, 2).contiguous().view(n_class, B // a_batch_size, T * a_batch_size).permute(1, 0, 2)
    return torch_ref.style_extractor(sub("style_extractor."), ci, cl, n_class=n_class)


@pytest.mark.parametrize("use_pred", [True, False])
def test_get_style_matches_oracle(cuda, tmp_path, use_pred):
    """generate.get_style (reference generate.py:48-83, style_together): HIP recogniser + alignment + author collapse + style extractor vs the
    oracle composition on the same seeded weights"""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.generate import get_style
    from handwriting_line_generation_amd.harness import build_gan_trainer
    from handwriting_line_generation_amd.model import HWWithStyle
    from oracle import cases, torch_ref
    import json, os
    cfg_model = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "model_config_iam.json")))
    sd = torch_ref.seeded_state_dict(HWWithStyle(cfg_model), 21)
    trainer, cfg = build_gan_trainer("iam_gan", 2, 2, width=192, label_len=8, workdir=str(tmp_path), model_state=sd)
    cfg["trainer"]["style_together"] = True
    cfg["trainer"]["use_hwr_pred_for_style"] = use_pred
    g = torch.Generator().manual_seed(6)
    image = torch.rand(4, 1, 64, 192, generator=g) * 2 - 1
    label = torch.randint(1, 80, (8, 4), generator=g, dtype=torch.int32)
    inst = {"image": image, "label": label, "a_batch_size": 2}
    rng.set_mode("host")
    try:
        trainer.model.train()          # the recogniser normalises with batch statistics in training mode, as in the reference's trainer
        with torch.no_grad():
            style = get_style(cfg, trainer.model, inst, trainer.gpu)
        ref = _oracle_style(sd, image, label.long(), 2, use_pred)
    finally:
        rng.set_mode("device")
    assert style.shape == ref.shape == (2, 128)
    err = float((style.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-4, err


def test_forward_generate_interpolate_match_oracle(cuda, tmp_path):
    """model(label, lengths, style) end to end under the reference's host RNG (numpy draws in insert_spaces, torch noise in the generator):
    spacer -> insert_spaces -> one-hot -> generator vs the oracle; generate() and interpolate() (generate.py:796-828) on top of it"""
    import json, os
    import torch.nn.functional as F
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.generate import generate, interpolate
    from handwriting_line_generation_amd.harness import build_gan_trainer, CHAR_FILES
    from handwriting_line_generation_amd.model import HWWithStyle
    from handwriting_line_generation_amd.utils import string_utils
    from oracle import seq_oracle, torch_ref
    cfg_model = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "model_config_iam.json")))
    sd = torch_ref.seeded_state_dict(HWWithStyle(cfg_model), 21)
    # a spacer that spreads the text out (the seeded one predicts ~0 blanks, a line would be only a few columns wide)
    sd["spacer.mean"] = torch.tensor([[3.0, 1.0]]).view_as(sd["spacer.mean"])
    trainer, cfg = build_gan_trainer("iam_gan", 1, 1, width=128, label_len=6, workdir=str(tmp_path), model_state=sd)
    model = trainer.model
    char_to_idx = json.load(open(CHAR_FILES["iam"]))["char_to_idx"]
    text = "hello world"
    g = torch.Generator().manual_seed(2)
    s1, s2 = torch.randn(1, 128, generator=g), torch.randn(1, 128, generator=g)
    sub = lambda p: {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}   # noqa: E731

    def oracle_line(style):
        lab = torch.from_numpy(string_utils.str2label_single(text, char_to_idx).astype(np.int64))[:, None]
        counts = torch_ref.spacer(sub("spacer."), F.one_hot(lab, 80).float(), style, training=False)
        spaced, _ = seq_oracle.insert_spaces(lab, [lab.shape[0]], counts, 80, 1e-8, 1e-9)
        return torch_ref.generator(sub("generator."), spaced, style)
    rng.set_mode("host")
    model.eval()
    try:
        with torch.no_grad():
            torch.manual_seed(4); np.random.seed(4)
            img = generate(model, s1.to(cuda), text, char_to_idx, cuda)
            torch.manual_seed(4); np.random.seed(4)
            ref = oracle_line(s1)
            assert img.shape == ref.shape and img.shape[3] > 4 * len(text)
            assert float((img.cpu() - ref).abs().max()) < 1e-4
            torch.manual_seed(5); np.random.seed(5)
            imgs, styles = interpolate(model, s1.to(cuda), s2.to(cuda), text, char_to_idx, cuda, step=0.25)
            assert len(imgs) == len(styles) == 4
            torch.manual_seed(5); np.random.seed(5)
            for k, alpha in enumerate(np.arange(0, 1.0, 0.25)):
                st = s2 * float(alpha) + float(1 - alpha) * s1
                assert float((styles[k] - st).abs().max()) < 1e-6
                r = oracle_line(st)
                assert imgs[k].shape == r.shape and float((imgs[k].cpu() - r).abs().max()) < 1e-4, (k, float((imgs[k].cpu() - r).abs().max()))
    finally:
        rng.set_mode("device")
        model.train()


def test_large_generation_batches_take_the_banked_style_path(cuda, tmp_path):
    """forward-only calls run the style MLP (hwg_mlp_chain_fwd) and the ten AdaIN affines (hwg_linear_bank_fwd) as one launch each for ANY
    batch (row chunks of 16 per block); with gradients enabled batches above 16 lines take the per-layer path. Same generator output either
    way (different summation order in the style path: within the 1e-4 of BASELINE's north star on the tanh image, and no worse in the later
    row chunks than in the first), for a batch that is not a multiple of the chunk."""
    from handwriting_line_generation_amd import ops, rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    torch.manual_seed(0)
    trainer, cfg = build_gan_trainer("iam_gan", 1, 2, width=256, label_len=12, workdir=str(tmp_path))
    gen = trainer.model.generator
    gen.eval()
    g = torch.Generator().manual_seed(4)
    B, T = 40, 23
    idx = torch.randint(0, cfg["model"]["num_class"], (T, B), generator=g)
    content = torch.nn.functional.one_hot(idx, cfg["model"]["num_class"]).float().to(trainer.gpu)
    style = torch.randn(B, cfg["model"]["style_dim"], generator=g).to(trainer.gpu)
    calls = []
    orig = ops.L.call

    def call(fn, *a):
        calls.append(fn)
        return orig(fn, *a)
    ops.L.call = call
    try:
        rng.set_mode("device", seed=21)
        with torch.no_grad():
            banked = gen(content, style)
        n_banked = (calls.count("hwg_linear_bank_fwd"), calls.count("hwg_mlp_chain_fwd"))
        del calls[:]
        rng.set_mode("device", seed=21)
        layered = gen(content, style).detach()
        n_layered = (calls.count("hwg_linear_bank_fwd"), calls.count("hwg_mlp_chain_fwd"))
    finally:
        ops.L.call = orig
        rng.set_mode("device")
    assert n_banked == (1, 1) and n_layered == (0, 0), (n_banked, n_layered)
    assert banked.shape == layered.shape == (B, 1, 64, 4 * T)
    err = (banked - layered).abs().amax(dim=(1, 2, 3))
    assert float(err.max()) < 1e-4, err
    assert float(err[16:].max()) < 4 * float(err[:16].max()) + 1e-6, err


def test_forward_only_generator_draws_its_noise_in_the_kernel_bit_identically(cuda, tmp_path):
    """device generator, no gradients: rng.NoiseBlock hands out VirtualNoise records and ops.adain_epilogue draws the normals inside its kernel
    (hwg_adain_fwd_rng) - the same Philox counters, the same Box-Muller arithmetic (csrc/philox.h) as the one hwg_randn launch of a training
    pass. The image must be BIT-identical to the pass that materialises the noise, the stream must advance by the same amount, and no
    hwg_randn may run."""
    from handwriting_line_generation_amd import ops, rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    torch.manual_seed(0)
    trainer, cfg = build_gan_trainer("iam_gan", 1, 2, width=256, label_len=12, workdir=str(tmp_path))
    gen = trainer.model.generator
    gen.eval()
    g = torch.Generator().manual_seed(6)
    B, T = 5, 19
    idx = torch.randint(0, cfg["model"]["num_class"], (T, B), generator=g)
    content = torch.nn.functional.one_hot(idx, cfg["model"]["num_class"]).float().to(trainer.gpu)
    style = torch.randn(B, cfg["model"]["style_dim"], generator=g).to(trainer.gpu)
    calls = []
    orig = ops.L.call

    def call(fn, *a):
        calls.append(fn)
        return orig(fn, *a)
    ops.L.call = call
    try:
        rng.set_mode("device", seed=33)
        with torch.no_grad():
            virt = gen(content, style)
        off_virtual = rng.device_rng().offset
        n_virtual = (calls.count("hwg_randn"), calls.count("hwg_adain_fwd_rng"), calls.count("hwg_adain_fwd"))
        del calls[:]
        rng.set_mode("device", seed=33)
        real = gen(content, style).detach()            # gradients enabled: the noise tensors exist (one hwg_randn), hwg_adain_fwd reads them
        off_real = rng.device_rng().offset
        n_real = (calls.count("hwg_randn"), calls.count("hwg_adain_fwd_rng"), calls.count("hwg_adain_fwd"))
        # a TAPED forward (the GAN trainer's balanced lessons) runs under no_grad as well, but its backward pass reads the noise and every
        # epilogue must be on the tape: not a forward-only pass
        del calls[:]
        rng.set_mode("device", seed=33)
        gen.tape_mode = True
        try:
            taped = gen(content, style.clone().requires_grad_(True))
            tapes = gen.take_tapes()
        finally:
            gen.tape_mode = False
        n_taped = (calls.count("hwg_randn"), calls.count("hwg_adain_fwd_rng"), calls.count("hwg_adain_fwd"))
    finally:
        ops.L.call = orig
        rng.set_mode("device")
    assert n_virtual == (0, 10, 0) and n_real == (1, 0, 10), (n_virtual, n_real)
    assert n_taped == (1, 0, 10) and len(tapes) == 1 and torch.equal(taped.detach(), real), n_taped
    assert sum(1 for nd in tapes[0].tape.nodes if getattr(nd[0], "__name__", "") == "_AdaIN") == 10
    assert off_virtual == off_real > 0
    assert torch.equal(virt, real)
    # a VirtualNoise record materialises to the very tensor the training pass would have read
    v = ops.VirtualNoise(33, 7, (2, 3, 5, 4))
    want = torch.empty(2, 3, 5, 4, device=trainer.gpu)
    ops.L.call("hwg_randn", want, want.numel(), 33, 7, torch.cuda.current_stream().cuda_stream)
    assert torch.equal(v.materialise(trainer.gpu), want)
