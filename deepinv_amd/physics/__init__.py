"""Operator API mirror of ``deepinv.physics`` for the accelerated hot path."""
from .forward import (Physics, LinearPhysics, DecomposablePhysics, Denoising, adjoint_function, power_method)
from .noise import NoiseModel, ZeroNoise, GaussianNoise
from .mri import MRI, MultiCoilMRI, MRIMixin
from .tomography import Tomography, RampFilter
from .blur import Blur, BlurFFT, Downsampling
from . import functional
from . import generator
