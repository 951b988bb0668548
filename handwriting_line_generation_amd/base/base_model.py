"""Minimal counterpart of the reference's base/base_model.py: a torch.nn.Module that remembers its config."""
import logging

import numpy as np
from torch import nn


class BaseModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.logger = logging.getLogger(self.__class__.__name__)

    def summary(self):
        n = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        self.logger.info("Trainable parameters: %d", n)
        self.logger.info(self)
