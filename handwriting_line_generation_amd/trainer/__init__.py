from .hw_with_style_trainer import HWWithStyleTrainer  # noqa: F401
from .auto_trainer import AutoTrainer  # noqa: F401
