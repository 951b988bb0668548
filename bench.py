#!/usr/bin/env python
"""bench.py - G+D train steps/sec of the GAN curriculum (BASELINE.json metric) on N MI355X GPUs of one node.

  python bench.py --gpus 1 --steps 14 --warmup 7
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one `HWWithStyleTrainer._train_iteration` (one lesson of the 7-lesson curriculum: count / gen / auto / disc ...)
on one rank's synthetic author batch; steps are timed in whole curriculum cycles where possible. Data parallel runs give every rank
its own author shard (weak scaling) and all-reduce the gradient sets over RCCL; `value` = N*K / max-over-ranks elapsed time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # per-GPU shape of BASELINE.json configs[3] (DDP: 4 authors per GPU, a_batch_size 2 -> 8 lines of 64x512), used at every N
    "iam_gan_b4a2_w512": dict(which="iam_gan", batch_size=4, a_batch_size=2, width=512, label_len=30),
    # BASELINE.json configs[2] exactly as SURVEY 8d reads it (one author, one line per step)
    "iam_gan_b1a1_w512": dict(which="iam_gan", batch_size=1, a_batch_size=1, width=512, label_len=30),
    # the shipped config's own batch (2 authors x 2 lines) - the shape of the survey's CPU measurement
    "iam_gan_b2a2_w512": dict(which="iam_gan", batch_size=2, a_batch_size=2, width=512, label_len=30),
    # BASELINE.json configs[4]: RIMES, variable widths 256..1024 padded per batch
    "rimes_gan_b4a2_w256_1024": dict(which="rimes_gan", batch_size=4, a_batch_size=2, width=1024, min_width=256, label_len=40),
    # SURVEY 8d "peaked recogniser": the default workload with a trained-looking recogniser output (70 % blanks, confident characters), so
    # that the per-character expert bank of the style extractor runs a realistic number of windows / experts
    "iam_gan_b4a2_w512_peaked": dict(which="iam_gan", batch_size=4, a_batch_size=2, width=512, label_len=30, peaked=True),
    # BASELINE.json configs[1]: the style autoencoder's pre-training step (AutoTrainer: Encoder2 -> DecoderNoSkip + E_HWR, L1 + CTC), the
    # shipped batch of 28 lines of 64x512. Not a GAN curriculum: every step is the same lesson; reported as "AE train steps/sec"
    "iam_auto_b28_w512": dict(which="iam_auto", batch_size=28, a_batch_size=1, width=512, label_len=30),
}
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 at 256 CU x 2.4 GHz
LESSONS = ("count", "gen", "auto", "disc", "gen", "auto", "disc")   # the shipped GAN curriculum


def peaked_offset_fn(num_class, seed=5, blank_frac=0.7, gain=10.0):
    """logit offset [B,1,T,C] that makes a randomly initialised recogniser predict like a trained one: per column a fixed class (blank with
    probability 0.7) gets +10. The pattern differs per line (rolled by 7 columns per line index)."""
    cache = {}

    def fn(B, T, C, device):
        key = (B, T, C)
        if key not in cache:
            g = torch.Generator().manual_seed(seed)
            cls = torch.randint(1, num_class, (4096,), generator=g)
            cls[torch.rand(4096, generator=g) < blank_frac] = 0
            off = torch.zeros(B, 1, T, C)
            for b in range(B):
                idx = cls[(torch.arange(T) + 7 * b) % 4096]
                off[b, 0, torch.arange(T), idx] = gain
            cache[key] = off.to(device)
        return cache[key]
    return fn


class ClockSampler:
    """GPU clock / power state during the timed region, so that a box-to-box difference can be read from the record: a thread samples the
    amdgpu sysfs files of the device (current sclk / mclk level, hwmon power) every 100 ms - file reads, no GPU calls, no subprocess inside
    the timed region."""

    def __init__(self, index):
        import glob
        self.files = {}
        self.samples = {}
        self.note = None
        try:
            bus = torch.cuda.get_device_properties(index)
            want = None
            for attr in ("pci_bus_id", "pci_device_id", "pci_domain_id"):
                if not hasattr(bus, attr):
                    want = None
                    break
            else:
                want = "%04x:%02x:%02x.0" % (bus.pci_domain_id, bus.pci_bus_id, bus.pci_device_id)
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
            cards = [c for c in cards if os.path.exists(os.path.join(c, "pp_dpm_sclk"))]
            pick = None
            for c in cards:
                if want and os.path.realpath(c).endswith(want):
                    pick = c
            if pick is None and cards:
                pick = cards[min(index, len(cards) - 1)]
                self.note = "device matched by index, not by PCI address"
            if pick:
                self.files["sclk_mhz"] = os.path.join(pick, "pp_dpm_sclk")
                self.files["mclk_mhz"] = os.path.join(pick, "pp_dpm_mclk")
                for h in glob.glob(os.path.join(pick, "hwmon", "hwmon*")):
                    for name in ("power1_average", "power1_input"):
                        if os.path.exists(os.path.join(h, name)):
                            self.files["power_w"] = os.path.join(h, name)
                            break
                    if os.path.exists(os.path.join(h, "freq1_input")):
                        self.files["sclk_mhz"] = os.path.join(h, "freq1_input")
                self.card = pick
        except Exception as e:  # noqa: BLE001 - a missing sysfs node must never cost the bench line
            self.note = "sysfs probe failed: %r" % (e,)
        self._stop = False
        self._thread = None

    @staticmethod
    def _read(key, path):
        txt = open(path).read()
        if path.endswith(("pp_dpm_sclk", "pp_dpm_mclk")):
            for line in txt.splitlines():          # "1: 2100Mhz *"
                if line.rstrip().endswith("*"):
                    return float(line.split(":")[1].strip().split("M")[0])
            return None
        v = float(txt.strip())
        if key == "power_w":
            return v / 1e6
        if path.endswith("freq1_input"):
            return v / 1e6
        return v

    def _run(self):
        while not self._stop:
            for k, p in self.files.items():
                try:
                    v = self._read(k, p)
                    if v is not None:
                        self.samples.setdefault(k, []).append(v)
                except Exception:  # noqa: BLE001
                    pass
            time.sleep(0.1)

    def start(self):
        import threading
        if self.files:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread is not None:
            self._thread.join(timeout=1.0)
        out = {"source": "amdgpu sysfs, sampled every 100 ms inside the timed region" if self.files else None, "note": self.note}
        for k, v in self.samples.items():
            if v:
                out[k] = {"min": round(min(v), 1), "mean": round(sum(v) / len(v), 1), "max": round(max(v), 1), "samples": len(v)}
        if not self.samples:
            # no readable sysfs nodes for this user: fall back to one rocm-smi query right after the timed region (the GPU is idle by then,
            # so this is the idle state - said so in the record)
            try:
                import subprocess
                r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
                out["rocm_smi_after_timed_region_idle"] = json.loads(r.stdout) if r.stdout.strip().startswith("{") else r.stdout[-400:]
            except Exception as e:  # noqa: BLE001
                out["rocm_smi_after_timed_region_idle"] = "unavailable: %r" % (e,)
        return out


def rccl_summary(path):
    """what RCCL reported at communicator init (NCCL_DEBUG=INFO, INIT + GRAPH subsystems, written to `path`): rank count, library version,
    number of channels, the transports of the ring / tree connections - enough to see from the JSON line whether an N-GPU run really ran
    N ranks over xGMI P2P"""
    if not path or not os.path.exists(path):
        return None
    import re
    text = open(path, errors="replace").read()
    out = {"log": path}
    m = re.search(r"nranks (\d+)", text)
    if m:
        out["nranks"] = int(m.group(1))
    m = re.search(r"(RCCL|NCCL) version ([\w.+-]+)", text)
    if m:
        out["version"] = m.group(1) + " " + m.group(2)
    ch = re.findall(r"(\d+) coll channels", text)
    if ch:
        out["coll_channels"] = int(ch[-1])
    out["via"] = sorted(set(re.findall(r"via (P2P/[\w/]+|SHM[\w/]*|NET/[\w/]+|direct[\w/]*)", text)))[:6]
    out["rings_trees"] = {"ring_lines": len(re.findall(r"Ring \d+ :", text)), "tree_lines": len(re.findall(r"Trees? ", text))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=140)
    ap.add_argument("--warmup", type=int, default=14)
    ap.add_argument("--workload", default="iam_gan_b4a2_w512", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gen", action="store_true", help="skip the secondary gen lines/sec measurement")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of CPU work for the oracle baseline")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the short child-process runs of the other BASELINE configs (iam_auto_b28_w512, iam_gan_b1a1_w512, rimes_gan_b4a2_w256_1024) "
                         "that follow the timed region of the default workload")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible and there is no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("HWG_DIST_BACKEND", "nccl")   # "gloo" lets two ranks share one GPU for a functional check of the DP path
    if backend == "nccl" and world > ndev:
        raise SystemExit("%d ranks but only %d GPUs visible" % (world, ndev))
    local = local % ndev
    torch.cuda.set_device(local)
    force_dp = bool(int(os.environ.get("HWG_FORCE_DP", "0") or 0))     # one-rank process group with the data-parallel exchange switched on
    rccl_log = None
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl" and "NCCL_DEBUG" not in os.environ:
            # self-diagnosing multi-GPU runs: RCCL's init / topology report goes to a per-rank file (stdout must stay one JSON line) and
            # rank 0 folds its summary (ranks, channels, transports) into the JSON's data_parallel.rccl
            import tempfile
            rccl_log = os.path.join(tempfile.gettempdir(), "hwg_rccl_rank%d_%d.log" % (rank, os.getpid()))
            os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH", NCCL_DEBUG_FILE=rccl_log)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)

    # The step's host side is thousands of tiny torch-CPU / numpy ops; torch's default intra-op pool (one thread per visible core, 256 on
    # the GPU box against a 16-CPU quota) only adds wake-up latency to them: 40.5 -> 42.8 steps/s with a single thread.
    torch.set_num_threads(1)
    from handwriting_line_generation_amd import ops, rng
    from handwriting_line_generation_amd.harness import build_gan_trainer

    wl = WORKLOADS[args.workload]
    # The contract: W untimed warm-up steps, then EXACTLY K timed steps - `steps` / `warmup` in the JSON are the driver's numbers. The curriculum
    # is a 7-lesson cycle whose lessons differ 6x in cost (auto 25 ms, disc 4.5 ms), and the launch-list recorder (replay.py) needs its
    # sightings before the step rate is the product's steady state; both are dealt with BEFORE the counted warm-up, in a recorder warm-up whose
    # length is reported (`recorder_warmup_steps`): whole cycles until the recorder is quiet, plus 0-6 lessons that put the K-step window at
    # the curriculum phase whose lesson mix is closest to the cycle's mean cost (`timed_window`; irrelevant when K is a multiple of 7).
    steps_req, warm_req = args.steps, args.warmup
    args.warmup = max(args.warmup, 0)
    args.steps = max(args.steps, 1)
    torch.manual_seed(1234 + rank); np.random.seed(1234 + rank); random.seed(1234 + rank)
    rng.set_mode("device", seed=99 + rank)
    # identical initial weights on every rank (seeded init before the rank-dependent seeds matter): build under a fixed seed
    torch.manual_seed(0)
    gan = wl["which"].endswith("_gan")
    if gan:
        trainer, cfg = build_gan_trainer(wl["which"], wl["batch_size"], wl["a_batch_size"], width=wl["width"], label_len=wl["label_len"],
                                         min_width=wl.get("min_width"), gpu=local, rank=rank, world=world)
    else:
        from handwriting_line_generation_amd.harness import build_simple_trainer
        trainer, cfg = build_simple_trainer(wl["which"], batch_size=wl["batch_size"], width=wl["width"], label_len=wl["label_len"], gpu=local,
                                            rank=rank, world=world)
        trainer.flush_log = lambda: {}
        args.no_gen = True
    torch.manual_seed(1234 + rank)
    if wl.get("peaked"):
        trainer.model.hwr.logit_offset = peaked_offset_fn(cfg["model"]["num_class"])
    if world > 1:
        for p in trainer.model.parameters():
            dist.broadcast(p.data, 0)
        for b in trainer.model.buffers():
            dist.broadcast(b.data, 0)

    # launch-list replay of the frozen recogniser's passes (handwriting_line_generation_amd/replay.py): host side only, results bit-identical to
    # the eager path (self-checked per recorded geometry; tests/test_trainer_gpu.py); HWG_REPLAY=0 keeps every pass eager
    from handwriting_line_generation_amd import replay as _replay
    _replay.enable()
    # inputs resident in HBM before the timed region (the contract's "inputs already resident"): a ring of synthetic batches built up
    # front (the loader wraps around; the text lessons draw from the corpus on the host as the reference does)
    trainer.data_loader.make_resident(min(args.warmup + args.steps + 35, 192), trainer.gpu)
    trainer.data_loader_iter = iter(trainer.data_loader)
    # losses of step i are read back while steps i+1 .. i+lag run (every read-back still happens inside the timed region: flush_log() before
    # the closing barrier). lag 1 pins the host to the GPU once per lesson, which starves the short gen / disc lessons (host enqueue time
    # ~ GPU time there, tools/host_time.py) of the lead the host gains during the long auto lessons; lag 2 lets that lead carry over.
    trainer.async_log = int(os.environ.get("HWG_LOG_LAG", "2"))
    cycle = 7 if gan else 1   # lessons of the shipped GAN curriculum: count, gen, auto, disc, gen, auto, disc

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def agree_max(v):
        """the same number on every rank (loop counts must agree: every rank runs the same lessons)"""
        if world > 1:
            t = torch.tensor([v], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return v

    # weight gradients on a second stream: "auto" = in the auto lessons only (GPU-bound: three backward passes over every network), 1 = in
    # every lesson, 0 = never
    sw = os.environ.get("HWG_SIDE_WGRAD", "auto")
    side_wgrad = "auto" if sw == "auto" else bool(int(sw or 0))
    if not gan:
        side_wgrad = False
    trainer._side_wgrad = side_wgrad
    ops.SIDE_WGRAD = side_wgrad is True

    # Recorder warm-up (before the counted warm-up, not part of `warmup`): whole cycles for at least WARM_SECONDS of wall time (clock / power
    # state) and until the launch-list recorder (replay.py) has gone quiet: a pass is recorded at its 3rd / 24th sighting (a one-off eager run
    # of every backward variant plus a self-check, 20-50 ms), i.e. around cycle 12 for the recogniser on generated lines. Warm until no program
    # has been recorded for QUIET_CYCLES cycles, at least REPLAY_MIN_CYCLES and at most 40 cycles. `recorder_warmup_steps` reports its length.
    WARM_SECONDS = 1.5
    it = 0
    barrier()
    tw = time.perf_counter()
    pre_done = 0
    QUIET_CYCLES, REPLAY_MIN_CYCLES = 4, 14
    quiet, recorded = 0, -1
    pre_marks = None
    while True:
        last = [torch.cuda.Event(enable_timing=True) for _ in range(cycle + 1)]
        for k in range(cycle):
            last[k].record()
            trainer._train_iteration(it); it += 1
        last[cycle].record()
        pre_marks = last
        pre_done += cycle
        now = _replay.STATS["captures"] + _replay.STATS["rejected"]
        quiet, recorded = (quiet + 1 if now == recorded else 0), now
        torch.cuda.synchronize()
        settled = (not _replay.ENABLED) or not gan or (quiet >= QUIET_CYCLES and pre_done >= REPLAY_MIN_CYCLES * cycle) or pre_done >= 40 * cycle
        waiting = 0.0 if (time.perf_counter() - tw >= WARM_SECONDS and settled) else 1.0
        if agree_max(waiting) == 0.0 or pre_done >= 100 * cycle:
            break
    # phase of the timed window: with K not a multiple of the cycle the window's lesson mix depends on where it starts; pick the start whose
    # mix costs closest to K x the cycle's mean (lesson costs = GPU time between step starts in the last recorder warm-up cycle)
    timed_window = None
    if cycle > 1:
        cost = [pre_marks[k].elapsed_time(pre_marks[k + 1]) for k in range(cycle)]       # index = lesson (the loop above runs whole cycles from lesson 0)
        mean = sum(cost) / cycle
        best, best_err = 0, None
        for s0 in range(cycle):
            err = abs(sum(cost[(s0 + k) % cycle] for k in range(args.steps)) - args.steps * mean)
            if best_err is None or err < best_err - 1e-9:
                best, best_err = s0, err
        best = int(agree_max(float(best)))
        extra = (best - args.warmup - it) % cycle
        for _ in range(extra):
            trainer._train_iteration(it); it += 1
        pre_done += extra
        timed_window = {"first_lesson": "%d:%s" % (best, LESSONS[best]), "lesson_mix_cost_vs_cycle_mean": round(1.0 + (sum(cost[(best + k) % cycle] for k in range(args.steps)) - args.steps * mean) / (args.steps * mean), 4),
                        "how": "start phase of the K timed steps chosen before the counted warm-up so that the window's lesson mix costs closest to K x the cycle mean (exact when K % 7 == 0)"}
    # the counted warm-up: exactly W untimed steps
    for _ in range(args.warmup):
        trainer._train_iteration(it); it += 1

    from handwriting_line_generation_amd.model.char_style import CharStyleEncoder
    from handwriting_line_generation_amd.trainer import flat_params
    CharStyleEncoder.stats.update(calls=0, windows=0, experts=0)
    flat_params.COMM.update(collectives=0, bytes=0)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-lesson GPU time: one event between steps
    first_lesson = it % cycle
    clocks = ClockSampler(local)
    clocks.start()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        marks[k].record()
        trainer._train_iteration(it); it += 1
    marks[args.steps].record()
    trainer.flush_log()
    barrier()
    elapsed = time.perf_counter() - t0
    clock_report = clocks.stop()
    st = dict(CharStyleEncoder.stats)
    comm = dict(flat_params.COMM)

    # Secondary: whole curriculum cycles (>= 10 cycles and >= 1.5 s) right after the timed region - the figure earlier rounds reported as `value`
    # when they rounded the driver's K up to whole cycles; kept so that a K that is not a multiple of 7 can be compared with the cycle average
    whole = None
    if gan:
        n_w = int(agree_max(float(max(10 * cycle, int(1.5 / max(elapsed / args.steps, 1e-6)) // cycle * cycle + cycle))))
        n_w = min(n_w, 2100)
        barrier()
        tq = time.perf_counter()
        for _ in range(n_w):
            trainer._train_iteration(it); it += 1
        trainer.flush_log()
        barrier()
        eq = agree_max(time.perf_counter() - tq)
        whole = {"value": round(world * n_w / eq, 4), "unit": "steps/s", "steps": n_w, "ms_per_step": round(eq / n_w * 1e3, 3),
                 "what": "whole curriculum cycles timed the same way right after the K timed steps; not part of `value`"}

    # Roofline measurement, AFTER the timed region (round 3 profiled cycles inside it: `value` carried the event packets and the
    # side-stream-off mode on a third of the driver's steps). The library records every MFMA launch's own begin / end timestamps
    # (hwg_prof_*); the weight-gradient side stream is off here because co-running kernels inflate each other's durations.
    profiling = rank == 0 and not os.environ.get("HWG_BENCH_NO_PROF")
    PROF_CYCLES = int(os.environ.get("HWG_BENCH_PROF_CYCLES", "2"))
    prof_steps = 0
    prof_elapsed = 0.0
    if world > 1:
        profiling_any = True   # every rank runs the same lessons (collectives inside); only rank 0 records
    else:
        profiling_any = profiling
    prof = []
    if profiling_any and not os.environ.get("HWG_BENCH_NO_PROF"):
        if profiling:
            ops.prof_start()
            ops.prof_enable(True)
        trainer._side_wgrad = False
        ops.SIDE_WGRAD = False
        conc = getattr(trainer, "_concurrent_style_passes", None)
        if conc is not None:
            trainer._concurrent_style_passes = False     # (same reason: the per-group style passes share the chip when they run on their own streams)
        barrier()
        tp = time.perf_counter()
        for _ in range(PROF_CYCLES * cycle):
            trainer._train_iteration(it); it += 1
            prof_steps += 1
        trainer.flush_log()
        barrier()
        prof_elapsed = time.perf_counter() - tp
        if profiling:
            ops.prof_enable(False)
            prof = ops.prof_stop()
        trainer._side_wgrad = side_wgrad
        ops.SIDE_WGRAD = side_wgrad is True
        if conc is not None:
            trainer._concurrent_style_passes = conc

    # Secondary figure, AFTER the timed region that `value` reports: the same loop with dead-gradient elimination on (trainer.skip_unused_grads:
    # no weight-gradient kernels for the frozen recogniser and for the discriminator outside disc lessons - SURVEY 8d's "minimum necessary"
    # variant; weights and losses are bit-identical, tests/test_trainer_gpu.py). `value` itself is the reference's launches, as executed.
    min_nec = None
    if gan and not os.environ.get("HWG_BENCH_NO_MINNEC"):
        trainer.skip_unused_grads = True
        n2 = 3 * cycle
        # untimed: switch the gradient requirements / plans over and let the launch-list recorder settle again - the recogniser's backward passes
        # without weight gradients are other programs (recorded at their 3rd / 24th sighting, 20-50 ms each: inside a 21-step window they read
        # as 83 steps/s where the settled loop runs 110). Same rule as the recorder warm-up above, bounded.
        quiet2, seen2, cycles2 = 0, -1, 0
        while cycles2 < 20:
            for _ in range(cycle):
                trainer._train_iteration(it); it += 1
            cycles2 += 1
            now2 = _replay.STATS["captures"] + _replay.STATS["rejected"]
            quiet2, seen2 = (quiet2 + 1 if now2 == seen2 else 0), now2
            settled2 = (not _replay.ENABLED) or (quiet2 >= QUIET_CYCLES and cycles2 >= REPLAY_MIN_CYCLES)
            if agree_max(0.0 if settled2 else 1.0) == 0.0:
                break
        trainer.flush_log()
        barrier()
        t1 = time.perf_counter()
        for _ in range(n2):
            trainer._train_iteration(it); it += 1
        trainer.flush_log()
        barrier()
        e2 = agree_max(time.perf_counter() - t1)
        trainer.skip_unused_grads = False
        min_nec = {"value": round(world * n2 / e2, 4), "unit": "steps/s", "steps": n2, "ms_per_step": round(e2 / n2 * 1e3, 3),
                   "what": "same loop with trainer.skip_unused_grads (no weight gradients for the frozen recogniser / for the discriminator outside disc "
                           "lessons); measured after the timed region and after the recorder has settled on the new programs, not part of `value`",
                   "settle_cycles": cycles2}
    elapsed = agree_max(elapsed)
    lesson_ms = {}
    for k in range(args.steps):
        name = "%d:%s" % ((first_lesson + k) % cycle, LESSONS[(first_lesson + k) % cycle] if gan else "auto-pretrain")
        lesson_ms.setdefault(name, []).append(marks[k].elapsed_time(marks[k + 1]))
    per_lesson_ms = {n: round(sum(v) / len(v), 3) for n, v in sorted(lesson_ms.items())}

    # secondary metric of BASELINE.json ("and gen lines/sec"): HWWithStyle.forward(label, lengths, style) -> images, inference only,
    # measured AFTER the timed training region on the same weights (spacer -> host insert_spaces -> generator; one line = 64 x ~4T px)
    gen = None
    if rank == 0 and not args.no_gen:
        model = trainer.model
        model.eval()
        B = 64            # a generation service batches requests; the training batch (8 lines) is launch-bound at 1.6 ms per call
        gsteps = 48
        g = torch.Generator().manual_seed(4321)
        labels = [ops.h2d(torch.randint(1, cfg["model"]["num_class"], (wl["label_len"], B), generator=g, dtype=torch.int32), trainer.gpu) for _ in range(8)]
        lengths = torch.IntTensor([wl["label_len"]] * B)
        styles = [ops.h2d(torch.randn(B, cfg["model"]["style_dim"], generator=g), trainer.gpu) for _ in range(8)]
        from handwriting_line_generation_amd.generate import generate_stream
        labels_host = [l.cpu() for l in labels]
        reqs = lambda n: ((labels_host[i % 8], lengths, styles[i % 8]) for i in range(n))   # noqa: E731
        with torch.no_grad():
            for img, _ in generate_stream(model, reqs(5)):
                pass
            torch.cuda.synchronize()
            tg = time.perf_counter()
            px = 0
            for img, _ in generate_stream(model, reqs(gsteps)):     # spacer of request i+1 overlaps the host step of request i
                px += img.shape[3]
            torch.cuda.synchronize()
            tg = time.perf_counter() - tg
        model.train()
        gen = {"value": round(B * gsteps / tg, 1), "unit": "lines/s (per GPU, inference)", "lines_per_call": B, "chars_per_line": wl["label_len"],
               "mean_line_width_px": round(px / gsteps, 1), "ms_per_call": round(tg / gsteps * 1e3, 3)}

    if rank == 0:
        fam = {}
        by_shape = {}
        for kind, shape, flops, dt in prof:
            f = fam.setdefault(kind, [0.0, 0.0, 0])
            f[0] += flops; f[1] += dt; f[2] += 1
            g = by_shape.setdefault((kind, shape), [0.0, 0.0, 0])
            g[0] += flops; g[1] += dt; g[2] += 1
        if os.environ.get("HWG_CONV_DUMP"):
            with open(os.environ["HWG_CONV_DUMP"], "w") as fh:
                fh.write("# per-shape MFMA conv launches in the profiled cycles after the timed region (%d steps): time_ms  launches  avg_us  TFLOP/s  kind  shape(N,H,W,C,K,R,S,stride,pad,dil,mode,network)  work (FLOPs; bytes for the reduce kinds)\n" % prof_steps)
                for (kind, shape), (fl, sec, n) in sorted(by_shape.items(), key=lambda kv: -kv[1][1]):
                    fh.write("%9.3f %6d %9.1f %7.1f  %s %s  %.6e\n" % (sec * 1e3, n, sec / n * 1e6, fl / sec / 1e12 if sec > 0 else 0, kind, shape, fl))
        # the north-star target is stated on the G+D conv stack: every conv kernel (MFMA, direct, their reduce passes) launched by the
        # generator's or the discriminator's layers, forward and backward, algorithmic FLOPs / kernel time
        gd = {"G": [0.0, 0.0], "D": [0.0, 0.0]}
        by_net = {}
        for kind, shape, work, dt in prof:
            sc = shape[-1] if shape else None
            bn = by_net.setdefault(str(sc), [0.0, 0.0])
            bn[1] += dt
            if "reduce" not in kind:
                bn[0] += work
            if sc in gd:
                gd[sc][1] += dt
                if "reduce" not in kind:
                    gd[sc][0] += work
        mf = {k: v for k, v in fam.items() if "reduce" not in k and "direct" not in k}
        dom = max(mf, key=lambda k: mf[k][1]) if mf else None
        roofline = None
        if dom:
            fl, sec, n = fam[dom]
            ach = fl / sec / 1e12
            issue = 2.25 if dom.startswith("wino") else 1.0     # Winograd F(2x2,3x3) / F(3x3,2x2): 16 multiplies issued per 36 counted
            roofline = {"bound": "mfma", "kernel": dom, "achieved": round(ach / issue, 3), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        # `frac` = what the matrix cores issued / peak (never > 1); `frac_direct_form` = the layer's direct-form FLOPs (SURVEY 8d:
                        # 2 x MACs of the convolution) / time / peak = effective throughput, which exceeds the issued rate by the Winograd saving
                        "frac": round(ach / issue / PEAK_FP32_MFMA_TFLOPS, 4), "achieved_direct_form": round(ach, 3),
                        "frac_direct_form": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "issued_over_direct_form": round(1.0 / issue, 4),
                        "traffic": None, "launches": n, "avg_launch_us": round(sec / n * 1e6, 2),
                        "gflop_per_launch": round(fl / n / 1e9, 4),
                        "other_kernels": {k: {("achieved_GBps" if k.endswith("reduce_kernel") else "achieved"): round(v[0] / v[1] / (1e9 if k.endswith("reduce_kernel") else 1e12), 3),
                                              "launches": v[2], "time_frac_of_step": round(v[1] / max(prof_elapsed, 1e-9), 3)}
                                          for k, v in fam.items() if k != dom},
                        "time_frac_of_step": round(sec / max(prof_elapsed, 1e-9), 3),
                        "profiled_steps": prof_steps, "profiled_where": "whole curriculum cycles run after the timed region (kernel begin / end timestamps "
                                                                          "of every launch; weight-gradient side stream off)"}
            if dom.startswith("wino"):
                # Winograd F(2x2,3x3) / F(3x3,2x2) kernels are credited with the layer's direct-form FLOPs (FLOPs = 2 MAC of the
                # convolution) but issue 16 instead of 36 multiplies per 2x2 tile and channel pair: the matrix cores are busy for 1/2.25 of it
                roofline["flops_counted"] = ("achieved / frac: MFMA FLOPs issued (direct-form / 2.25); achieved_direct_form / frac_direct_form: 2 x MACs of the "
                                             "convolution, the algorithmic work of SURVEY 8d")
                roofline["mfma_issued_frac"] = round(ach / 2.25 / PEAK_FP32_MFMA_TFLOPS, 4)
            for k, v in roofline["other_kernels"].items():
                if k.startswith("wino") and "achieved" in v:
                    v["achieved_direct_form"] = v.pop("achieved")
                    v["achieved"] = round(v["achieved_direct_form"] / 2.25, 3)
                    v["mfma_issued_frac"] = round(v["achieved"] / PEAK_FP32_MFMA_TFLOPS, 4)
            gfl, gsec = gd["G"][0] + gd["D"][0], gd["G"][1] + gd["D"][1]
            if gsec > 0:
                # FLOPs are those of the launches made in the profiled cycles = the reference's launches, as executed (231.9 GFLOP per
                # step at 8 lines; the discriminator's weight gradients of gen / auto lessons are computed here as there). 3x3 layers on the
                # Winograd kernels count their direct-form FLOPs; time = every conv kernel of the two networks incl. their reduce passes.
                # tools/prof_summary.py gd <HWG_CONV_DUMP file> recomputes these figures from the per-shape dump of the same run.
                roofline["gd_conv_stack"] = {"flop_variant": "as executed by this implementation (the reference's launches, discriminator weight gradients of gen / auto lessons included)",
                                             "achieved": round(gfl / gsec / 1e12, 3), "frac": round(gfl / gsec / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                                             "gflop_per_step": round(gfl / 1e9 / max(prof_steps, 1), 1), "ms_per_step": round(gsec * 1e3 / max(prof_steps, 1), 3),
                                             "generator": round(gd["G"][0] / gd["G"][1] / 1e12, 3) if gd["G"][1] > 0 else None,
                                             "discriminator": round(gd["D"][0] / gd["D"][1] / 1e12, 3) if gd["D"][1] > 0 else None}
            roofline["conv_time_by_network_ms_per_step"] = {k: [round(v[1] * 1e3 / max(prof_steps, 1), 3), round(v[0] / v[1] / 1e12, 1) if v[1] > 0 else None]
                                                            for k, v in sorted(by_net.items(), key=lambda kv: -kv[1][1])}
        if roofline:
            # HBM bytes per launch of the dominant kernel: PMC passes (FETCH_SIZE / WRITE_SIZE, collected separately with rocprofv3 on this
            # same command and corrected per MI355X_MICROARCH.md) are committed under profiles/; they cannot be collected from inside bench.py
            import hashlib
            for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                try:
                    raw = open(os.path.join(ROOT, "profiles", name), "rb").read()
                    pm = json.loads(raw)
                    if pm.get("workload") == args.workload and dom in pm["kernels"]:
                        roofline["traffic"] = pm["kernels"][dom]["hbm_bytes_per_launch_corrected"]
                        roofline["traffic_source"] = "profiles/%s (rocprofv3 PMC over the whole step, bytes per launch)" % name
                        roofline["traffic_source_sha256"] = hashlib.sha256(raw).hexdigest()
                        # PMC counters need rocprofv3 around the process (separate --pmc passes, MI355X_MICROARCH.md): they cannot be collected from
                        # inside the timed process, so this is a committed pass of the same command, identified by its hash
                        roofline["traffic_measured_in_run"] = False
                        break
                except (OSError, KeyError, ValueError):
                    pass
            # ... and per layer shape (tools/pmc_shapes.py replays the step's top conv shapes under the same counters): measured HBM bytes against
            # the one-pass operand bytes 4*(input + weights + output), weighted by this run's launch counts
            try:
                shapes_file = next((f for f in ("r06_pmc_shapes.json", "r05_pmc_shapes.json", "r04_pmc_shapes.json", "r03_pmc_shapes.json", "r02_pmc_shapes.json") if os.path.exists(os.path.join(ROOT, "profiles", f))), "r02_pmc_shapes.json")
                ps = json.load(open(os.path.join(ROOT, "profiles", shapes_file)))
                launches = {(k, repr(sh)): v[2] for (k, sh), v in by_shape.items()}
                tot = {}
                for row in ps["shapes"]:
                    n_here = launches.get((row["kind"], row["shape"]), 0)
                    t = tot.setdefault(row["kind"], [0.0, 0.0, 0])
                    t[0] += n_here * (row["hbm_fetch_bytes"] + row["hbm_write_bytes"]); t[1] += n_here * row["algorithmic_bytes"]; t[2] += n_here
                roofline["traffic_by_shape"] = {"source": "profiles/" + shapes_file,
                                                "kernels": {k: {"launches_covered": t[2], "launches": fam[k][2], "hbm_bytes_per_launch": round(t[0] / t[2]),
                                                                "operand_bytes_per_launch": round(t[1] / t[2]), "ratio": round(t[0] / t[1], 2)}
                                                            for k, t in tot.items() if t[2] > 0 and k in fam}}
            except (OSError, KeyError, ValueError):
                pass
        cpu = None
        if world == 1 and not args.no_cpu_baseline and gan:
            # the oracle's CPU cycle runs in a child process (it never touches the GPU) under a hard timeout
            import subprocess
            B = wl["batch_size"] * wl["a_batch_size"]
            cmd = [sys.executable, "-m", "oracle.cycle_ref", str(B), str(wl["a_batch_size"]), str(wl["width"]), str(wl["label_len"]), str(args.cpu_budget)]
            try:
                r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=max(240.0, 10 * args.cpu_budget))
                j = json.loads(r.stdout.strip().splitlines()[-1])
                cpu_model = None
                try:
                    cpu_model = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.lower().startswith("model name")), None)
                except OSError:
                    pass
                cpu = {"value": round(j["steps_per_sec"], 4), "unit": "steps/s", "cores": j["cores"], "kind": "port", "cpu_model": cpu_model,
                       "sample": "%d steps (whole 7-lesson cycles) of oracle/cycle_ref.py, torch %s fp32 CPU, same batch shape (%d lines 64x%d), %.1f s"
                                 % (j["steps"], j["torch"], B, wl["width"], j["seconds"])}
            except Exception as e:  # noqa: BLE001 - the baseline is a reported extra, never a reason to lose the GPU number
                cpu = {"value": None, "unit": "steps/s", "cores": None, "kind": "port", "sample": "cpu baseline failed: %r" % (e,)}
        # The other BASELINE configs, as short child-process runs of this same file AFTER everything above (a child is a fresh process: no
        # exec of a process that has touched the GPU; the parent's few GB stay resident next to it): driver-visible numbers for configs[1],
        # [2] and [4] instead of builder-only ones. Fixed budget: 42 / 60 timed steps each, no CPU baseline, no generation figure.
        others = None
        if world == 1 and args.workload == "iam_gan_b4a2_w512" and not args.no_other_workloads and not os.environ.get("HWG_BENCH_NO_OTHERS"):
            import subprocess
            others = {}
            for wname, k_steps in (("iam_auto_b28_w512", 60), ("iam_gan_b1a1_w512", 42), ("rimes_gan_b4a2_w256_1024", 42)):
                env = dict(os.environ, HWG_BENCH_NO_MINNEC="1", HWG_BENCH_PROF_CYCLES="1")
                env.pop("HWG_CONV_DUMP", None)
                t_child = time.perf_counter()
                try:
                    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", wname, "--steps", str(k_steps), "--warmup", "7", "--no-cpu-baseline",
                                        "--no-gen", "--no-other-workloads"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=240.0)
                    j = json.loads(r.stdout.strip().splitlines()[-1])
                    rl = j.get("roofline") or {}
                    others[wname] = {"metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "steps": j["steps"], "warmup": j["warmup"],
                                     "recorder_warmup_steps": j.get("recorder_warmup_steps"), "whole_cycles": (j.get("whole_cycles") or {}).get("value"),
                                     "roofline_frac": rl.get("frac"), "roofline_frac_direct_form": rl.get("frac_direct_form"),
                                     "gd_conv_stack": {"frac": (rl.get("gd_conv_stack") or {}).get("frac"), "achieved": (rl.get("gd_conv_stack") or {}).get("achieved")},
                                     "replay": j.get("replay"), "config": j.get("config"), "wall_s": round(time.perf_counter() - t_child, 1)}
                except Exception as e:  # noqa: BLE001 - an extra must never cost the headline line
                    others[wname] = {"value": None, "error": repr(e)[:300], "wall_s": round(time.perf_counter() - t_child, 1)}
        out = {
            "metric": "G+D train steps/sec" if gan else "AE train steps/sec", "value": round(world * args.steps / elapsed, 4), "unit": "steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "steps_requested": steps_req, "warmup_requested": warm_req,
            "recorder_warmup_steps": pre_done, "timed_window": timed_window, "whole_cycles": whole, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "config_file": cfg["name"], "lines_per_gpu_step": wl["batch_size"] * wl["a_batch_size"],
                       "authors_per_gpu": wl["batch_size"], "a_batch_size": wl["a_batch_size"], "line_px": "64x%d" % wl["width"],
                       "curriculum": "count,gen,auto,disc,gen,auto,disc" if gan else None, "parallelism": "dp%d" % world},
            "roofline": roofline, "gen_lines_per_sec": gen, "cpu_baseline": cpu, "minimum_necessary": min_nec, "other_workloads": others,
            # GPU time between the starts of consecutive steps (HIP events on the step stream), averaged per lesson of the curriculum
            "per_lesson_ms": per_lesson_ms,
            # data parallel: ranks in the process group and this rank's all-reduce traffic (gradient sets + None-masks) per step
            "data_parallel": {"world_size": world, "backend": (dist.get_backend() if dist.is_initialized() else None), "forced_single_rank_exchange": force_dp,
                              "rccl": rccl_summary(rccl_log),
                              "collectives_per_step": round(comm["collectives"] / args.steps, 2),
                              "allreduce_mbytes_per_step": round(comm["bytes"] / args.steps / 1e6, 2)},
            "replay": (lambda r: dict(r.STATS, enabled=bool(r.ENABLED)))(__import__("handwriting_line_generation_amd.replay", fromlist=["STATS"])),
            "side_stream_wgrad": side_wgrad,   # off in the roofline-profiled cycles, which run after the timed region
            "log_lag": int(trainer.async_log),
            "concurrent_style_passes": getattr(trainer, "_concurrent_style_passes", None),   # likewise off in the profiled cycles
            # sclk / mclk / socket power sampled from sysfs every 100 ms during the timed region (min / mean / max), and rocm-smi's view
            "clocks": clock_report,
            "hbm_peak_mb": round(max(torch.cuda.max_memory_allocated(), getattr(trainer, "hbm_peak_bytes", 0)) / 1e6, 1),   # peak of torch's allocator on rank 0 (weights, optimizer state, flat gradient sets, activations, arenas)
            "inputs_resident": True,     # one synthetic batch per step built and uploaded before the timed region (SyntheticLoader.make_resident)
            # load of the per-character expert bank (K18): style extractions in the timed region, character windows and distinct experts per call
            "style_extractor_load": {"recogniser": "peaked (70% blanks, +10 logit on one class per column)" if wl.get("peaked") else "random-init on uniform-noise lines",
                                     "calls": st["calls"], "windows_per_call": round(st["windows"] / max(st["calls"], 1), 1),
                                     "experts_present_per_call": round(st["experts"] / max(st["calls"], 1), 1)},
        }
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
