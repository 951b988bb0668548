#!/usr/bin/env python
"""python train.py -c configs/<cfg>.json [-r checkpoint] [-s soft_checkpoint] [-g gpu]
Drop-in for the reference's train.py: classes are resolved by the names in the config (arch / loss / trainer.class), the
config's `name` must match its file name, SIGINT saves a checkpoint. Under torchrun (WORLD_SIZE>1) every rank trains on its
own author shard and gradients are all-reduced over RCCL. Real datasets (IAM/RIMES images) are outside the accelerated path;
`--synthetic` drives the trainer with synthetic author batches of the configured shape."""
import argparse
import json
import logging
import os
import signal
import sys

import torch

torch.set_num_threads(1)   # host side = many tiny CPU ops; the intra-op pool only adds latency (see bench.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
logging.basicConfig(level=logging.INFO, format="")


def main():
    ap = argparse.ArgumentParser(description="MI355X-native handwriting GAN trainer")
    ap.add_argument("-c", "--config", type=str)
    ap.add_argument("-r", "--resume", type=str, default=None)
    ap.add_argument("-s", "--soft_resume", type=str, default=None)
    ap.add_argument("-g", "--gpu", type=int, default=None)
    ap.add_argument("--synthetic", action="store_true", help="synthetic author batches instead of a dataset on disk")
    ap.add_argument("--iterations", type=int, default=None)
    ap.add_argument("--random-init-aux", action="store_true",
                    help="allow a RANDOM-INIT perceptual encoder / recogniser when trainer.encoder_weights / model.pretrained_hwr do not exist "
                         "(the reference fails there); implied by --synthetic")
    ap.add_argument("--reference-gradients", action="store_true",
                    help="also compute the parameter gradients the reference computes and nothing reads (the frozen recogniser's; the "
                         "discriminator's in gen / auto lessons): the reference's launches, same weights and losses to the bit, ~20 %% slower")
    args = ap.parse_args()

    resume = args.resume
    if resume is None and args.soft_resume is not None and os.path.exists(args.soft_resume):
        resume = args.soft_resume
    if args.config is not None:
        config = json.load(open(args.config))
        if config["name"] != os.path.basename(args.config)[3:-5]:
            raise SystemExit("config name %r does not match its file name" % config["name"])
    elif resume is not None:
        from handwriting_line_generation_amd.logger import load_checkpoint
        config = load_checkpoint(resume)["config"]
    else:
        raise SystemExit("need -c or -r")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        config["gpu"] = local
    elif args.gpu is not None:
        config["gpu"] = args.gpu
    if args.iterations is not None:
        config["trainer"]["iterations"] = args.iterations
    # dead-gradient elimination is the drop-in's default: losses, updates and every weight are bit-identical to the loop
    # that computes them (tests/test_trainer_gpu.py::test_skip_unused_grads_changes_no_weight_and_no_loss); a config key or the flag decide otherwise
    if args.reference_gradients:
        if int(config["trainer"].get("skip_unused_grads", 0) or 0):
            print("train.py: --reference-gradients overrides trainer.skip_unused_grads=%r of the config" % config["trainer"]["skip_unused_grads"], flush=True)
        config["trainer"]["skip_unused_grads"] = 0          # the explicit flag wins over the config key
    else:
        config["trainer"].setdefault("skip_unused_grads", 1)
    if rank == 0:
        print("train.py: skip_unused_grads = %d (%s)" % (int(config["trainer"]["skip_unused_grads"] or 0),
              "dead-gradient elimination: the reference's never-read parameter gradients are not computed; weights and losses bit-identical"
              if int(config["trainer"]["skip_unused_grads"] or 0) else "the reference's launches, including the gradients nothing reads"), flush=True)
    # the frozen recogniser's passes as recorded launch lists (replay.py; bit-identical, self-checked; HWG_REPLAY=0 turns it off)
    from handwriting_line_generation_amd import replay as _replay
    _replay.enable()

    # Random streams: one base seed per run (config["seed"] or drawn here and shared by all ranks), rank r draws from stream
    # (base, r): generator noise and Dropout2d masks differ between data-parallel ranks and between runs. The Philox offset is
    # stored in every checkpoint, so -r / -s continue the stream instead of replaying it.
    from handwriting_line_generation_amd import rng
    base_seed = config.get("seed")
    if base_seed is None:
        base_seed = int.from_bytes(os.urandom(4), "little")
        if world > 1:
            t = torch.tensor([base_seed], dtype=torch.int64, device="cuda")
            torch.distributed.broadcast(t, 0)
            base_seed = int(t.item())
    rng.seed_process(base_seed, rank)

    import handwriting_line_generation_amd.model as models
    import handwriting_line_generation_amd.model.loss as losses
    import handwriting_line_generation_amd.trainer as trainers
    from handwriting_line_generation_amd.data.synthetic import SyntheticAuthorDataset, SyntheticLoader, write_synthetic_corpus

    dl = config["data_loader"]
    pkg_data = os.path.join(ROOT, "handwriting_line_generation_amd", "data")
    if not os.path.exists(dl["char_file"]):
        dl["char_file"] = os.path.join(pkg_data, os.path.basename(dl["char_file"]))
    # The reference fails when the warm-start files are missing (model/hw_with_style.py:166-178 loads `pretrained_hwr`,
    # trainer/hw_with_style_trainer.py:136-160 loads `encoder_weights`). A GAN trained against a random perceptual encoder or a random
    # frozen recogniser is a different experiment, so stand-ins are written only on request and announced loudly.
    tr = config["trainer"]
    aux_ok = args.synthetic or args.random_init_aux
    if "encoder_weights" in tr and not os.path.exists(tr["encoder_weights"]):
        if not aux_ok:
            raise SystemExit("trainer.encoder_weights %r does not exist (train the autoencoder config first; --random-init-aux substitutes a "
                             "RANDOM-INIT encoder)" % tr["encoder_weights"])
        logging.getLogger("train").warning("WARNING: %r missing -> the perceptual loss uses a RANDOM-INIT Encoder2 (--%s)",
                                           tr["encoder_weights"], "synthetic" if args.synthetic else "random-init-aux")
        os.makedirs(os.path.dirname(tr["encoder_weights"]) or ".", exist_ok=True)
        if rank == 0:
            torch.save({"state_dict": models.Autoencoder({"type": tr.get("encoder_type", "2tight"), "hwr": config["model"]["num_class"]}).state_dict()},
                       tr["encoder_weights"])
    if config["model"].get("pretrained_hwr") and not os.path.exists(config["model"]["pretrained_hwr"]):
        if not aux_ok:
            raise SystemExit("model.pretrained_hwr %r does not exist (train the recogniser config first; --random-init-aux keeps a RANDOM-INIT "
                             "frozen recogniser)" % config["model"]["pretrained_hwr"])
        logging.getLogger("train").warning("WARNING: %r missing -> the frozen recogniser stays RANDOM-INIT (--%s)",
                                           config["model"]["pretrained_hwr"], "synthetic" if args.synthetic else "random-init-aux")
        config["model"]["pretrained_hwr"] = None
    valid_loader = None
    if args.synthetic:
        ds = SyntheticAuthorDataset(dl["char_file"], dl["batch_size"], dl.get("a_batch_size", 1), width=512, label_len=30, num_batches=10 ** 9)
        loader = SyntheticLoader(ds, rank, world)
    elif os.path.isdir(os.path.join(dl["data_dir"], "xmls")) or os.path.exists(os.path.join(dl["data_dir"], "lines_training_2011.xml")):
        # a dataset on disk in the reference's layout (IAM: forms/ + xmls/ + sets.json; RIMES: images_gray/ + lines_*_2011*.xml):
        # author-grouped batches, every rank its own authors (data/author_hw_dataset.py, data/author_rimeslines_dataset.py)
        from handwriting_line_generation_amd.data.author_hw_dataset import getDataLoader
        loader, valid_loader = getDataLoader(config, "train", rank, world)
    else:
        raise SystemExit("no dataset at %r (expected the reference's layout: IAM forms/ xmls/ data/sets.json, or RIMES images_gray/ lines_training_2011.xml); pass --synthetic to train on "
                         "synthetic author batches of the configured shape" % dl["data_dir"])
    tr = config["trainer"]
    if "text_data" in tr and not os.path.exists(tr["text_data"]):
        os.makedirs(tr["save_dir"], exist_ok=True)
        tr["text_data"] = os.path.join(tr["save_dir"], "synthetic_corpus.txt")
        if rank == 0 and not os.path.exists(tr["text_data"]):
            write_synthetic_corpus(tr["text_data"], dl["char_file"])
    if world > 1:
        torch.distributed.barrier()

    model = getattr(models, config["arch"])(config["model"])
    loss = {k: getattr(losses, v) for k, v in config["loss"].items()} if isinstance(config["loss"], dict) else getattr(losses, config["loss"])
    from handwriting_line_generation_amd.logger import Logger
    trainer = getattr(trainers, config["trainer"]["class"])(model, loss, [], resume, config, loader, valid_loader, Logger())
    if world > 1:
        for p in list(trainer.model.parameters()) + list(trainer.model.buffers()):
            torch.distributed.broadcast(p.data, 0)

    # SIGINT -> checkpoint (reference train.py:72-75). The handler only raises a flag; BaseTrainer.train() looks at it between iterations,
    # ORs it over the ranks' control group and, when set anywhere, every rank calls save() at the same iteration and leaves. (Saving from
    # inside the handler would put rank 0 into _save_checkpoint's barrier alone, possibly between two asynchronous gradient all-reduces.)
    signal.signal(signal.SIGINT, lambda sig, frame: trainer.request_stop())
    trainer.train()


if __name__ == "__main__":
    main()
