from .hw_with_style_trainer import HWWithStyleTrainer  # noqa: F401
