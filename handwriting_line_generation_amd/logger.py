"""Training-history container stored in every checkpoint (reference: logger/logger.py:5-19).

Checkpoints written by the reference pickle an instance of `logger.logger.Logger`; `torch.load` can only unpickle them if
that dotted name resolves. `install_reference_aliases()` registers this module under the reference's module names (unless a
real `logger` package is importable already), and the class announces itself as `logger.logger.Logger`, so files saved here
load in the reference and vice versa.
"""
import json
import sys


class Logger:
    def __init__(self):
        self.entries = {}

    def add_entry(self, entry):
        self.entries[len(self.entries) + 1] = entry

    def __str__(self):
        return json.dumps(self.entries, sort_keys=True, indent=4)


Logger.__module__ = "logger.logger"


def install_reference_aliases():
    this = sys.modules[__name__]
    for name in ("logger", "logger.logger"):
        mod = sys.modules.get(name)
        if mod is None or not hasattr(mod, "Logger"):
            sys.modules[name] = this


def load_checkpoint(path):
    """torch.load of a checkpoint written by this package or by the reference (whose pickled `logger.logger.Logger` needs the aliases, and
    whose files torch >= 2.6 refuses under its weights-only default)"""
    import torch
    install_reference_aliases()
    return torch.load(path, map_location="cpu", weights_only=False)
