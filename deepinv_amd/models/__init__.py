from .base import Denoiser, Reconstructor
from .drunet import DRUNet
