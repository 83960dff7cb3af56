"""deepinv_amd — MI355X (gfx950) native forward/adjoint physics operators and the
PnP / unfolded iteration loop that calls them, behind deepinv's own Python API.

``import deepinv_amd as dinv`` then use ``dinv.physics.MRI``, ``dinv.optim.PGD`` … exactly as
with the reference.  All arithmetic of the operators runs in hand-written HIP kernels reached
through the C-ABI of ``libdeepinv_amd.so`` (``include/deepinv_amd.h``); there is no CPU path.
"""
__version__ = "0.1.0"

import os as _os

# HIP multiplexes the streams of a process onto GPU_MAX_HW_QUEUES hardware queues (4 by default); streams that share a queue
# run one after the other.  The batch lanes of the denoiser (models/drunet.py) need two streams that really overlap, also in a
# process whose RCCL communicator and graph capture hold streams of their own: ask for 8 queues unless the user decided
# otherwise.  Read by the HIP runtime when it initialises (the first device call of the process), so it has to be set before
# that - importing this package first is enough; hip.lane_streams() verifies the overlap either way.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
