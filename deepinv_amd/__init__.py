"""deepinv_amd — MI355X (gfx950) native forward/adjoint physics operators and the
PnP / unfolded iteration loop that calls them, behind deepinv's own Python API.

``import deepinv_amd as dinv`` then use ``dinv.physics.MRI``, ``dinv.optim.PGD`` … exactly as
with the reference.  All arithmetic of the operators runs in hand-written HIP kernels reached
through the C-ABI of ``libdeepinv_amd.so`` (``include/deepinv_amd.h``); there is no CPU path.
"""
__version__ = "0.1.0"

from . import hip  # noqa: F401
from . import physics  # noqa: F401
from . import models  # noqa: F401
from . import optim  # noqa: F401
from . import unfolded  # noqa: F401
from . import utils  # noqa: F401
from . import sampling  # noqa: F401
from . import distributed  # noqa: F401
from . import training  # noqa: F401
from .training import Trainer, train  # noqa: F401
