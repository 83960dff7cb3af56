from .potential import Potential
from .distance import Distance, L2Distance
from .data_fidelity import DataFidelity, L2, ZeroFidelity
from .prior import Prior, PnP, ZeroPrior
from .optim_iterators import (OptimIterator, fStep, gStep, PGDIteration, HQSIteration)
from .fixed_point import FixedPoint
from .optimizers import BaseOptim, PGD, HQS, optim_builder, create_iterator, BacktrackingConfig
from .linear import conjugate_gradient, least_squares, least_squares_implicit_backward, dot
from .linear_solvers import lsqr, bicgstab, minres
from .dpir import DPIR, get_DPIR_params
from . import linear
