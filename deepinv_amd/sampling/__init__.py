from .diffusion import DiffPIR
