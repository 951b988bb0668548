"""GPU: one full 7-lesson curriculum cycle of the HIP-backed HWWithStyleTrainer against the losses / parameter updates the
UNMODIFIED reference trainer produced on the same seeded weights, synthetic batches and RNG streams
(tests/golden/trainer_cycle.json, made by tools/gen_golden_trainer.py)."""
import json
import math
import os
import random

import numpy as np
import pytest
import torch

from oracle import torch_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_curriculum_cycle_matches_reference(cuda, tmp_path):
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    from handwriting_line_generation_amd.model import Autoencoder, HWWithStyle
    gold = json.load(open(os.path.join(GOLD, "trainer_cycle.json")))
    cfg_model = json.load(open(os.path.join(GOLD, "model_config_iam.json")))
    msd = torch_ref.seeded_state_dict(HWWithStyle(cfg_model), gold["wseed_model"])
    esd = torch_ref.seeded_state_dict(Autoencoder({"type": "2tight", "hwr": 80}), gold["wseed_enc"])
    rng.set_mode("host")
    try:
        trainer, cfg = build_gan_trainer("iam_gan", gold["batch_size"], gold["a_batch_size"], width=gold["W"], label_len=gold["label_len"],
                                         workdir=str(tmp_path), model_state=msd, encoder_state=esd)
        before = {k: v.detach().clone() for k, v in trainer.model.named_parameters()}
        torch.manual_seed(0); np.random.seed(0); random.seed(0)
        bad = []
        for it, ref in enumerate(gold["logs"]):
            log = trainer._train_iteration(it)
            assert set(log) == set(ref), "iteration %d logs %s vs reference %s" % (it, sorted(log), sorted(ref))
            for k, rv in ref.items():
                # Until the first balanced generator step (it 2) the two runs see bit-identical inputs and must agree to fp32 rounding.
                # After it, parameters differ by the fp32 conditioning of the recogniser/CTC backward (~1e-2 for the reference's own
                # arithmetic vs fp64, tests/test_pipeline_gpu.py) and the adversarial terms amplify that: only a coarse bound holds (a different
                # but equally valid summation order in one small kernel moves the iteration-4 adversarial loss by 5 %).
                tol = 2e-5 * max(abs(rv), 1e-3) if it < 3 else 1e-1 * max(abs(rv), 2e-2)
                if abs(log[k] - rv) > tol:
                    bad.append("it%d %s: %.6g vs %.6g" % (it, k, log[k], rv))
        delta = {}
        for k, p in trainer.model.named_parameters():
            top = k.split(".")[0]
            delta[top] = delta.get(top, 0.0) + (p.detach() - before[k]).double().abs().sum().item()
        for top, rv in gold["param_abs_delta"].items():
            if abs(delta[top] - rv) > 2e-2 * max(rv, 1e-9):
                bad.append("param |delta| %s: %.6g vs %.6g" % (top, delta[top], rv))
        sd = trainer.model.state_dict()
        for k, rv in gold["small_params_after"].items():
            got = sd[k].flatten().cpu().tolist()
            if max(abs(a - b) for a, b in zip(got, rv)) > 1e-4:
                bad.append("%s after cycle: %s vs %s" % (k, got, rv))
        assert not bad, "; ".join(bad)
    finally:
        rng.set_mode("device")


def test_skip_unused_grads_changes_no_weight_and_no_loss(cuda, tmp_path):
    """trainer.skip_unused_grads drops the weight-gradient kernels of parameters whose gradients nothing reads (frozen recogniser; discriminator
    outside disc lessons). Two curriculum cycles with the switch on and off from the same seeds (device Philox noise, same stream): every
    logged loss and every parameter / buffer after the run must be BIT-identical, and the skipped tensors must really be untouched."""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    outs = []
    for skip in (False, True):
        rng.set_mode("device", seed=11)
        torch.manual_seed(3); np.random.seed(3); random.seed(3)
        trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=str(tmp_path / ("s%d" % skip)))
        trainer.skip_unused_grads = skip
        torch.manual_seed(5); np.random.seed(5); random.seed(5)
        logs = [trainer._train_iteration(it) for it in range(14)]
        f = trainer.flat
        names = [n for n, _ in trainer.model.named_parameters()]
        touched = {names[pi]: bool(f.touched[k]) for k, pi in enumerate(f.order)}
        outs.append((logs, {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}, touched))
    (la, sa, ta), (lb, sb, tb) = outs
    for it, (a, b) in enumerate(zip(la, lb)):
        assert a == b, "iteration %d: %s vs %s" % (it, a, b)
    for k, v in sa.items():
        assert torch.equal(v, sb[k]), "%s differs with skip_unused_grads" % k
    assert any(ta[k] for k in ta if k.startswith("hwr.")) and not any(tb[k] for k in tb if k.startswith("hwr."))


def test_deferred_reduce_and_eager_reduce_train_to_the_same_bits(cuda, tmp_path):
    """trainer.defer_wgrad_reduce (default on): the partial-image sums of every backward pass in one table-driven launch. Two curriculum
    cycles with it on and off from the same seeds: every logged loss and every parameter / buffer must be BIT-identical, and the deferred
    run must really have gone through the multi-reduce launches."""
    from handwriting_line_generation_amd import ops, rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    outs = []
    for defer in (False, True):
        rng.set_mode("device", seed=11)
        torch.manual_seed(3); np.random.seed(3); random.seed(3)
        trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=str(tmp_path / ("d%d" % defer)))
        trainer._defer_reduce = defer
        torch.manual_seed(5); np.random.seed(5); random.seed(5)
        before, sn_before = ops._defer["launches"], ops._defer["sn_launches"]
        logs = [trainer._train_iteration(it) for it in range(14)]
        torch.cuda.synchronize()
        outs.append((logs, {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}, ops._defer["launches"] - before,
                     ops._defer["sn_launches"] - sn_before))
    (la, sa, na, sna), (lb, sb, nb, snb) = outs
    assert na == 0 and nb >= 14, (na, nb)
    # the discriminator's spectral-norm layers walked backward together (hwg_spectral_bwd_multi) in every backward pass that reached it
    assert sna == 0 and snb >= 10, (sna, snb)
    for it, (a, b) in enumerate(zip(la, lb)):
        assert a == b, "iteration %d: %s vs %s" % (it, a, b)
    for k, v in sa.items():
        assert torch.equal(v, sb[k]), "%s differs with the deferred reduce" % k


def test_fused_step_and_masked_zero_grad_train_to_the_same_bits(cuda, tmp_path):
    """Round 6: (i) clip_grad_value_ + the NaN check + Adam of a stepping lesson as ONE launch (HipAdam.step(clip): hwg_mt_clip_adam) against the
    three separate passes (trainer.fused_step = 0); (ii) zero_grad as a masked multi-tensor pass over the tensors that have a gradient against
    the fill of the whole group slice (FlatParams._zero_all). Two curriculum cycles each way from the same seeds: every logged loss, every
    parameter / buffer, every gradient left in the flat buffer and both Adam moments BIT-identical (trainer/hw_with_style_trainer.py:379-391);
    and after zero_grad of both optimizers nothing but zeros is left outside the 'rest' group (no writer bypasses the touched marks)."""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    from handwriting_line_generation_amd.trainer import flat_params
    outs = []
    for new in (False, True):
        rng.set_mode("device", seed=11)
        torch.manual_seed(3); np.random.seed(3); random.seed(3)
        trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=str(tmp_path / ("f%d" % new)))
        trainer._fused_step = new
        trainer.flat._zero_all = not new
        flat_params.BALANCE_SETS = new      # (iii) the balanced adds of all stashed sets in one pass (hwg_mt_abs_sum_sets / hwg_mt_axpy_sets) vs a launch per set
        torch.manual_seed(5); np.random.seed(5); random.seed(5)
        logs = [trainer._train_iteration(it) for it in range(14)]
        torch.cuda.synchronize()
        f = trainer.flat
        state = {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}
        extra = {"grad": f.flat_grad.clone(), "m": trainer.optimizer.exp_avg.clone(), "v": trainer.optimizer.exp_avg_sq.clone(),
                 "md": trainer.optimizer_discriminator.exp_avg.clone(), "vd": trainer.optimizer_discriminator.exp_avg_sq.clone(),
                 "flag": f._flag.clone()}
        trainer.optimizer.zero_grad(); trainer.optimizer_discriminator.zero_grad()
        a, b = f.group_range["rest"]
        end = int(f.offsets[a]) if b > a else f.total
        extra["left"] = float(f.flat_grad[:end].abs().max())
        outs.append((logs, state, extra))
    flat_params.BALANCE_SETS = True
    (la, sa, ea), (lb, sb, eb) = outs
    for it, (a, b) in enumerate(zip(la, lb)):
        assert a == b, "iteration %d: %s vs %s" % (it, a, b)
    for k, v in sa.items():
        assert torch.equal(v, sb[k]), "%s differs with the fused step" % k
    for k in ("grad", "m", "v", "md", "vd", "flag"):
        assert torch.equal(ea[k], eb[k]), k
    assert ea["left"] == 0.0 and eb["left"] == 0.0 and int(eb["flag"]) == 0


def test_pipelined_logging_returns_the_same_losses_n_iterations_later(cuda, tmp_path):
    """trainer.async_log = n: iteration i returns the losses of iteration i - n (the host may run n iterations ahead of the GPU), flush_log()
    resolves what is outstanding. Against synchronous logging from the same seeds: the same log dictionaries, shifted by n; the same weights."""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    runs = {}
    for lag in (0, 1, 3):
        rng.set_mode("device", seed=11)
        torch.manual_seed(3); np.random.seed(3); random.seed(3)
        trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=str(tmp_path / ("l%d" % lag)))
        trainer.async_log = lag
        torch.manual_seed(5); np.random.seed(5); random.seed(5)
        logs = [trainer._train_iteration(it) for it in range(9)]
        last = trainer.flush_log()
        assert trainer.flush_log() == {}                      # nothing outstanding after a flush
        torch.cuda.synchronize()
        runs[lag] = (logs, last, {k: v.detach().clone() for k, v in trainer.model.state_dict().items()})
    sync_logs = runs[0][0]
    assert all(l and "loss" in l for l in sync_logs) and runs[0][1] == {}
    for lag in (1, 3):
        logs, last, state = runs[lag]
        assert logs[:lag] == [{}] * lag
        assert logs[lag:] == sync_logs[:9 - lag], "lag %d: the pipelined logs are not the synchronous ones shifted" % lag
        assert last == sync_logs[-1]
        for k, v in runs[0][2].items():
            assert torch.equal(v, state[k]), "%s differs with async_log=%d" % (k, lag)


def test_nothing_accumulates_from_cycle_to_cycle(cuda, tmp_path):
    """long runs: the packed-weight cache must hold parameters only (under no_grad - taped forwards, the disc lessons' detached generator pass -
    every derived weight is a "leaf" too: caching those kept one temporary + its packed image alive per generator pass, 0.4 MB per step at
    the bench batch), and torch's allocated bytes at the same point of the curriculum cycle must not creep."""
    import gc
    from handwriting_line_generation_amd import ops, rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    rng.set_mode("device", seed=11)
    torch.manual_seed(3); np.random.seed(3); random.seed(3)
    trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=str(tmp_path))
    trainer.data_loader.make_resident(8, trainer.gpu)            # a ring of batches: the same shapes come round again
    trainer.data_loader_iter = iter(trainer.data_loader)

    def mark():
        trainer.flush_log(); torch.cuda.synchronize(); gc.collect()
        return len(ops._pack_cache), torch.cuda.memory_allocated()
    for it in range(28):
        trainer._train_iteration(it)
    n0, a0 = mark()
    for it in range(28, 84):
        trainer._train_iteration(it)
    n1, a1 = mark()
    assert all(isinstance(e[3], torch.nn.Parameter) for e in ops._pack_cache.values())
    assert n1 == n0, "packed-weight cache grew from %d to %d entries over 8 cycles" % (n0, n1)
    assert a1 - a0 < 16e6, "allocated bytes grew by %.1f MB over 8 cycles" % ((a1 - a0) / 1e6)


def test_allocator_trim_changes_nothing_but_the_cache(cuda, tmp_path):
    """trainer.alloc_trim_every: the periodic release of torch's cached blocks (long runs with per-batch line widths splinter the cache) fires
    when the cache holds more than twice the peak in use, and leaves losses and weights bit-identical"""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    outs = []
    for every in (0, 7):
        rng.set_mode("device", seed=11)
        torch.manual_seed(3); np.random.seed(3); random.seed(3)
        trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=str(tmp_path / ("t%d" % every)))
        trainer._trim_every = every
        torch.manual_seed(5); np.random.seed(5); random.seed(5)
        # a block that is free (cached) by the time of the first trim; sized against what earlier tests of this process still hold (module-level
        # caches keep their trainers alive): the trim fires when the cache exceeds twice the peak in use
        hog = torch.empty((16 << 30) + 3 * torch.cuda.memory_allocated(), dtype=torch.uint8, device=trainer.gpu)
        del hog
        torch.cuda.reset_peak_memory_stats()
        logs = [trainer._train_iteration(it) for it in range(15)]
        torch.cuda.synchronize()
        outs.append((logs, {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}, trainer.allocator_trims))
    (la, sa, ta), (lb, sb, tb) = outs
    assert ta == 0 and tb >= 1, (ta, tb)
    assert la == lb
    for k, v in sa.items():
        assert torch.equal(v, sb[k]), k


def test_batched_generator_backward_equals_one_pass_per_loss_group(cuda, tmp_path):
    """trainer.batch_gen_backward (default on): the two / three gradients a balanced lesson sends through the generator (reference trainer
    :300-338, one backward() per loss group) go through it in ONE pass, stacked along the batch axis, every group accumulating into its
    own buffer. Against the sequential passes from the same seeds: identical None-patterns, every balanced gradient (read where the
    reference clips) equal to fp32 reordering noise, identical losses in the first cycle."""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    runs = []
    for batched in (False, True):
        rng.set_mode("device", seed=11)
        torch.manual_seed(3); np.random.seed(3); random.seed(3)
        trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=str(tmp_path / ("b%d" % batched)))
        trainer._batch_gen_backward = batched
        torch.manual_seed(5); np.random.seed(5); random.seed(5)
        grads = {}

        def hook(it, trainer=trainer, grads=grads):
            f = trainer.flat
            pos_of = {id(f.params[pi]): k for k, pi in enumerate(f.order)}
            grads[it] = {n: (p.grad.detach().clone() if f.touched[pos_of[id(p)]] else None) for n, p in trainer.model.named_parameters()}
        trainer.pre_clip_hook = hook
        logs = [trainer._train_iteration(it) for it in range(7)]
        torch.cuda.synchronize()
        runs.append((logs, grads))
    (la, ga), (lb, gb) = runs
    assert set(ga) == set(gb) and len(ga) >= 3
    worst = 0.0
    for it in sorted(ga):
        # yardstick per sub-network: conv biases in front of a batch-statistics BatchNorm have an analytically zero gradient (both runs hold
        # rounding noise there), so a tensor's difference is measured against max(its own norm, 1e-3 x the largest norm in its network)
        top = {}
        for n, a in ga[it].items():
            if a is not None:
                top[n.split(".")[0]] = max(top.get(n.split(".")[0], 0.0), float(a.double().norm()))
        for n, a in ga[it].items():
            b = gb[it][n]
            assert (a is None) == (b is None), "iteration %d %s: gradient present in one run only" % (it, n)
            if a is None:
                continue
            na = max(float(a.double().norm()), 1e-3 * top[n.split(".")[0]])
            if na < 1e-12:
                assert float(b.abs().max()) < 1e-10, (it, n)
                continue
            e = float((a.double() - b.double()).norm()) / na
            worst = max(worst, e)
            # the first step's gradients are identical up to the schedule the planner picks for 2 / 3 x the batch (fp32 summation order);
            # later iterations start from weights that differ by those roundings through Adam's sign-like first steps
            assert e < (2e-4 if it <= 2 else 0.5), "iteration %d %s: relative difference %.2e" % (it, n, e)
    assert any(n.startswith("generator.") and v is not None for n, v in ga[2].items())
    for k in la[1]:
        assert abs(la[1][k] - lb[1][k]) <= 1e-5 * max(1.0, abs(la[1][k])), (k, la[1][k], lb[1][k])
    print("batched vs sequential generator backward: worst relative gradient difference %.2e" % worst)


def test_style_passes_on_streams_are_bit_identical_to_one_stream(cuda, tmp_path):
    """trainer.concurrent_style_passes (default on): the per-loss-group backward passes through the taped style extractor + recogniser run
    on streams of their own (same kernels, same per-set order, different buffers). Against the same passes one after the other on the main
    stream: every balanced gradient (read where the reference clips), every logged loss and every parameter after two curriculum cycles
    must be BIT-identical - a cross-stream race (scratch memory, allocator reuse, deferred sums) would show up here."""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    runs = []
    for conc in (False, True):
        rng.set_mode("device", seed=11)
        torch.manual_seed(3); np.random.seed(3); random.seed(3)
        trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=str(tmp_path / ("c%d" % conc)))
        trainer._concurrent_style_passes = conc
        torch.manual_seed(5); np.random.seed(5); random.seed(5)
        grads = {}

        def hook(it, trainer=trainer, grads=grads):
            f = trainer.flat
            grads[it] = (f.flat_grad.clone(), f.touched.copy())
        trainer.pre_clip_hook = hook
        logs = [trainer._train_iteration(it) for it in range(14)]
        torch.cuda.synchronize()
        runs.append((logs, grads, {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}, trainer._style_streams is not None))
    (la, ga, sa, ua), (lb, gb, sb, ub) = runs
    assert not ua and ub, "the concurrent run must really have used its streams"
    for it in sorted(ga):
        assert np.array_equal(ga[it][1], gb[it][1]), "iteration %d: None-pattern differs" % it
        assert torch.equal(ga[it][0], gb[it][0]), "iteration %d: balanced gradients differ (%d elements)" % (it, int((ga[it][0] != gb[it][0]).sum()))
    for it, (a, b) in enumerate(zip(la, lb)):
        assert a == b, "iteration %d: %s vs %s" % (it, a, b)
    for k, v in sa.items():
        assert torch.equal(v, sb[k]), "%s differs" % k


def test_masked_stash_holds_zeros_outside_its_mask(cuda):
    """FlatParams.stash() copies only the tensors that have a gradient; a pooled buffer must nevertheless hold ZEROS everywhere else,
    because data-parallel segment all-reduces sum whole sub-network ranges of it (trainer/flat_params.py: stale / dirty clearing).
    Stash with mask A, release, stash with a different mask B: the reused buffer is exactly zero outside B and equals the gradient inside B;
    same when the first stash's mask was widened in place (what the OR over the ranks does) before its release."""
    import numpy as np
    from handwriting_line_generation_amd.trainer.flat_params import FlatParams
    g = torch.Generator().manual_seed(3)
    shapes = [(7,), (33, 5), (130,), (4, 4, 3, 3), (70001,), (9,), (256, 17)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g).to(cuda)) for s in shapes]
    flat = FlatParams(params, {"main": params[:5], "disc": params[5:]}, names=["a.w%d" % i if i < 3 else "b.w%d" % i for i in range(len(params))])
    pos_of = {id(flat.params[pi]): k for k, pi in enumerate(flat.order)}

    def fill(which, seed):
        gg = torch.Generator().manual_seed(seed)
        want = {}
        for i in which:
            v = torch.randn(*shapes[i], generator=gg).to(cuda)
            params[i].grad.copy_(v)
            flat.touched[pos_of[id(params[i])]] = True
            want[i] = v
        return want

    def check(stash, want):
        buf, mask = stash[0], stash[1]
        expect = torch.zeros_like(buf)
        for i, v in want.items():
            k = pos_of[id(params[i])]
            assert mask[k]
            expect[int(flat.offsets[k]): int(flat.offsets[k]) + v.numel()] = v.flatten()
        assert int(mask.sum()) == len(want)
        assert torch.equal(buf, expect), "stash differs from (gradient inside the mask, zero outside): %d elements" % int((buf != expect).sum())
        assert float(flat.flat_grad.abs().max()) == 0.0, "the gradient buffer must be zero after a stash"

    A, B, C = [0, 2, 4, 6], [1, 2, 5], [3]
    want = fill(A, 1)
    s1 = flat.stash(); flat.touched[:] = False
    check(s1, want)
    flat.release(s1)
    want = fill(B, 2)
    s2 = flat.stash(); flat.touched[:] = False
    assert s2[0].data_ptr() == s1[0].data_ptr(), "the pooled buffer should have been reused"
    check(s2, want)
    # widen the mask in place before the release (a data-parallel OR marks tensors another rank touched; their slots may then be non-zero)
    k3 = pos_of[id(params[3])]
    s2[1][k3] = True
    s2[0][int(flat.offsets[k3]): int(flat.offsets[k3]) + 5] = 7.0          # what an all-reduce could have summed in
    flat.release(s2)
    want = fill(C + [0], 3)
    s3 = flat.stash(); flat.touched[:] = False
    check(s3, want)
    flat.release(s3)
    want = fill([5], 4)                                                     # a set that overlaps nothing of the previous one
    s4 = flat.stash(); flat.touched[:] = False
    check(s4, want)
    torch.cuda.synchronize()


def test_taped_lessons_run_at_a_batch_above_the_banked_backward_limit(cuda, tmp_path):
    """the style MLP chain / AdaIN affine bank hold at most 16 rows in their BACKWARD kernels. A taped generator forward runs under
    no_grad, which used to look like a forward-only call and put the banked ops on the tape at B = 17 - the batched backward then raised.
    With 17 lines per batch the balanced (taped) gen / auto lessons must train, on the per-layer path, with finite losses and gradients."""
    from handwriting_line_generation_amd import rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    rng.set_mode("device", seed=11)
    torch.manual_seed(3); np.random.seed(3); random.seed(3)
    trainer, _ = build_gan_trainer("iam_gan", 17, 1, width=128, label_len=6, workdir=str(tmp_path), curriculum=[["no-step", "gen"], ["auto", "auto-gen"]])
    assert trainer._batch_gen_backward and trainer.balance_loss
    for it in range(2):
        log = trainer._train_iteration(it)
        assert all(math.isfinite(float(v)) for v in log.values()), log
    torch.cuda.synchronize()
    g = trainer.flat.flat_grad
    assert bool(torch.isfinite(g).all())
    gen = trainer.model.generator
    assert all(p.grad is not None and float(p.grad.abs().sum()) > 0 for m in gen.style_emb if hasattr(m, "weight") for p in (m.weight,))


def test_discriminator_applied_twice_in_one_pass_deferred_equals_eager(cuda, tmp_path):
    """a lesson with both 'disc' and 'gen' applies the discriminator twice inside ONE backward pass, so every spectral-norm layer's
    deferred sigma-path backward is queued twice with the same destination; entries of one hwg_spectral_bwd_multi launch are added with plain
    read-modify-writes, so the two must go to consecutive launches. Deferred vs eager from the same seeds: bit-identical losses and weights."""
    from handwriting_line_generation_amd import ops, rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    outs = []
    for defer in (False, True):
        rng.set_mode("device", seed=11)
        torch.manual_seed(3); np.random.seed(3); random.seed(3)
        trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=str(tmp_path / ("d%d" % defer)), curriculum=[["disc", "gen"]])
        trainer._defer_reduce = defer
        torch.manual_seed(5); np.random.seed(5); random.seed(5)
        sn_before = ops._defer["sn_launches"]
        logs = [trainer._train_iteration(it) for it in range(3)]
        torch.cuda.synchronize()
        outs.append((logs, {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}, ops._defer["sn_launches"] - sn_before))
    (la, sa, sna), (lb, sb, snb) = outs
    assert sna == 0 and snb >= 6, (sna, snb)       # two passes (one per application of the discriminator) per flush
    for it, (a, b) in enumerate(zip(la, lb)):
        assert a == b, "iteration %d: %s vs %s" % (it, a, b)
    for k, v in sa.items():
        assert torch.equal(v, sb[k]), "%s differs with the deferred reduce" % k


def test_taped_lessons_make_no_untracked_torch_ops(cuda, tmp_path, monkeypatch):
    """the taped generator / style-extractor forwards of a whole curriculum cycle under the tape guard (HWG_TAPE_CHECK): every op on a taped
    tensor goes through an ops.Function or is a recognised reshape - nothing a gradient path could be lost in"""
    from handwriting_line_generation_amd import ops, rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    monkeypatch.setattr(ops, "TAPE_CHECK", True)
    rng.set_mode("device", seed=11)
    torch.manual_seed(3); np.random.seed(3); random.seed(3)
    trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=128, label_len=6, workdir=str(tmp_path))
    assert trainer._batch_gen_backward and trainer._tape_style
    for it in range(7):
        log = trainer._train_iteration(it)
        assert all(math.isfinite(float(v)) for v in log.values()), log
    torch.cuda.synchronize()


def test_recogniser_replay_trains_to_the_same_bits(cuda, tmp_path):
    """replay.py: the frozen recogniser's forward / backward as recorded launch lists replayed by ONE C call per pass. Two curriculum cycles +
    with it on and off from the same seeds: every logged loss and every parameter / buffer BIT-identical (the recorded programs passed their
    own bit-exact self-check, the trainer's gradient-set redirects, deferred sums, side stream and style tapes run through them), and the
    replay must really have carried the passes."""
    from handwriting_line_generation_amd import ops, replay, rng
    from handwriting_line_generation_amd.harness import build_gan_trainer
    outs = []
    for on in (False, True):
        replay.reset()
        replay.ENABLED = on
        after, replay.RECORD_AFTER, replay.RECORD_AFTER_DX = (replay.RECORD_AFTER, replay.RECORD_AFTER_DX), 2, 2
        for k in replay.STATS:
            replay.STATS[k] = 0
        try:
            rng.set_mode("device", seed=11)
            torch.manual_seed(3); np.random.seed(3); random.seed(3)
            trainer, _ = build_gan_trainer("iam_gan", 2, 2, width=256, label_len=12, workdir=str(tmp_path / ("r%d" % on)))
            torch.manual_seed(5); np.random.seed(5); random.seed(5)
            logs = [trainer._train_iteration(it) for it in range(21)]
            torch.cuda.synchronize()
            outs.append((logs, {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}, dict(replay.STATS)))
        finally:
            replay.ENABLED = False
            replay.RECORD_AFTER, replay.RECORD_AFTER_DX = after
    (la, sa, st_off), (lb, sb, st_on) = outs
    assert st_off["fwd"] == 0 and st_on["captures"] >= 1 and st_on["fwd"] >= 10 and st_on["bwd"] >= 10 and st_on["rejected"] == 0, st_on
    for it, (a, b) in enumerate(zip(la, lb)):
        assert a == b, "iteration %d: %s vs %s" % (it, a, b)
    for k, v in sa.items():
        assert torch.equal(v, sb[k]), "%s differs with the recogniser replayed" % k
