"""On-device measurement synthesis (csrc/random.hip) through the reference-shaped API: Gaussian noise and the Cartesian
MRI mask generators (reference deepinv/physics/noise.py:197-330, generator/mri.py:15-384).  The same invariants are
checked on the host emulation in tests/test_emu_random.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gaussian_noise_fused_kernel(dev):
    import deepinv_amd as dinv

    x = torch.rand(4, 2, 64, 64, device=dev)
    nm = dinv.physics.GaussianNoise(0.2)
    torch.manual_seed(0)
    y1 = nm(x)
    torch.manual_seed(0)
    y2 = nm(x)
    y3 = nm(x)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)          # torch.manual_seed keeps its meaning
    z = (y1 - x) / 0.2
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1) < 0.02
    # explicit generator + seed argument, per-sample sigma
    nm = dinv.physics.GaussianNoise(torch.tensor([0.0, 0.1, 0.2, 0.4]), rng=torch.Generator(dev))
    a, b = nm(x, seed=3), nm(x, seed=3)
    assert torch.equal(a, b) and float((a[0] - x[0]).abs().max()) == 0
    assert abs(float((a[3] - x[3]).std()) - 0.4) < 0.02
    # physics.forward = noise(A(x)) keeps working, on a masked MRI operator too
    phys = dinv.physics.MRI(mask=(torch.rand(64, 64) > 0.5).float(), img_size=(2, 64, 64), device=dev,
                            noise_model=dinv.physics.GaussianNoise(0.05))
    y = phys(x)
    assert y.shape == x.shape and torch.isfinite(y).all()


@pytest.mark.parametrize("cls", ["RandomMaskGenerator", "GaussianMaskGenerator", "EquispacedMaskGenerator"])
def test_mask_generators(dev, cls):
    import deepinv_amd as dinv

    G = getattr(dinv.physics.generator, cls)
    gen = G((2, 320, 320), acceleration=4, center_fraction=0.08, device=dev, rng=torch.Generator(dev).manual_seed(0))
    m = gen.step(batch_size=8)["mask"]
    assert m.shape == (8, 2, 320, 320) and m.device.type == "cuda"
    assert set(m.unique().tolist()) == {0.0, 1.0}
    lines = m[:, 0, 0]                                        # every row / channel carries the same columns
    assert torch.equal(m, lines[:, None, None, :].expand_as(m))
    n_center = int(0.08 * 320)
    if cls != "EquispacedMaskGenerator":
        assert torch.all(lines.sum(-1) == 320 // 4)           # n_center + n_lines = W / acceleration columns, always
        lo = 320 // 2 - n_center // 2
        assert torch.all(lines[:, lo:lo + n_center] == 1)
    else:
        assert torch.all((lines.sum(-1) - 80).abs() <= 3)
    assert len({tuple(r.tolist()) for r in lines.cpu()}) > 1   # random across the batch
    m2 = gen.step(batch_size=8, seed=5)["mask"]
    assert torch.equal(m2, gen.step(batch_size=8, seed=5)["mask"])
    # k-t masks and on-the-fly image size, as in the reference docstrings (mri.py:149-157)
    kt = G((2, 8, 64, 64), acceleration=8, center_fraction=0.04, device=dev).step(batch_size=1)["mask"]
    assert kt.shape == (1, 2, 8, 64, 64)
    assert gen.step(batch_size=0, img_size=(32, 48))["mask"].shape == (2, 32, 48)
    # the masks drive the MRI operator directly
    phys = dinv.physics.MRI(img_size=(2, 320, 320), device=dev)
    y = phys.A(torch.rand(8, 2, 320, 320, device=dev), mask=m)
    assert torch.equal(y == 0, m == 0) or float(((y == 0) != (m == 0)).float().mean()) < 1e-6
