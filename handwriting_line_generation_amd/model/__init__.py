from .hw_with_style import HWWithStyle, correct_pred  # noqa: F401
from .autoencoder import Autoencoder, Encoder2, DecoderNoSkip, E_HWR  # noqa: F401
from .pure_gen import SpacedGenerator  # noqa: F401
from .discriminator_ap import DiscriminatorAP  # noqa: F401
from .cnn_only_hwr import CNNOnlyHWR  # noqa: F401
from .char_style import CharStyleEncoder  # noqa: F401
from .count_cnn import CountCNN  # noqa: F401
