from .synthetic import radial_mask
