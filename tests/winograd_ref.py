"""fp64 helpers shared by the emulated and the hardware tests of the Winograd F(4x4,3x3) kernels (test infrastructure)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def winograd4_magnitude(x, w):
    """sum of the MAGNITUDES of everything an output pixel of Winograd F(4x4,3x3) adds up, in fp64:
    |A^T| [ sum_ci |G g G^T| . (|B^T| |d| |B|) ] |A| - the scale of the kernel's rounding errors (transform cancellations included)"""
    from deepinv_amd.hip.drunet import winograd4_matrices
    bt, G, at = (m.abs() for m in winograd4_matrices())
    B, cin, H, W = x.shape
    cout = w.shape[0]
    U = (winograd4_matrices()[1] @ w.double() @ winograd4_matrices()[1].t()).abs()         # [co, ci, 6, 6]
    xp = torch.nn.functional.pad(x.double().abs(), (1, 1, 1, 1))
    d = xp.unfold(2, 6, 4).unfold(3, 6, 4)                                                  # [B, ci, ty, tx, 6, 6]
    V = bt @ d @ bt.t()
    M = torch.einsum("ocij,bcyxij->boyxij", U, V)
    Y = at @ M @ at.t()                                                                     # [B, co, ty, tx, 4, 4]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, cout, H, W)
