"""DiffPIR diffusion-HQS sampler (reference deepinv/sampling/diffusion.py:227-513)."""
from __future__ import annotations

import torch

from ..models.base import Reconstructor


class DiffPIR(Reconstructor):
    def __init__(self, model, data_fidelity, sigma=0.05, max_iter=100, zeta=0.1, lambda_=7.0, verbose=False,
                 device="cpu"):
        super().__init__()
        self.model, self.data_fidelity = model, data_fidelity
        self.lambda_, self.max_iter, self.zeta, self.verbose, self.device = lambda_, max_iter, zeta, verbose, device
        self.beta_start, self.beta_end = 0.1 / 1000, 20 / 1000
        self.num_train_timesteps = 1000
        self.sigma = sigma
        (self.sqrt_1m_alphas_cumprod, self.reduced_alpha_cumprod, self.sqrt_alphas_cumprod,
         self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod, self.betas) = self.get_alpha_beta()
        self.rhos, self.sigmas, self.seq = self.get_noise_schedule(sigma=sigma)

    def get_alpha_beta(self):
        """diffusion.py:323-345"""
        betas = torch.linspace(self.beta_start, self.beta_end, self.num_train_timesteps, dtype=torch.float32,
                               device=self.device)
        ac = torch.cumprod(1.0 - betas, dim=0)
        s_ac, s_1m = torch.sqrt(ac), torch.sqrt(1.0 - ac)
        return s_1m, torch.div(s_1m, s_ac), s_ac, torch.sqrt(1.0 / ac), torch.sqrt(1.0 / ac - 1), betas

    def get_noise_schedule(self, sigma):
        """diffusion.py:347-375 (vectorised; same values)"""
        T = self.num_train_timesteps
        sigmas = torch.flip(self.reduced_alpha_cumprod, dims=(0,)).clone()
        sigma_ks = self.sqrt_1m_alphas_cumprod / self.sqrt_alphas_cumprod
        sig = sigma if not isinstance(sigma, torch.Tensor) else sigma.to(sigma_ks.device)
        rhos = self.lambda_ * sig ** 2 / sigma_ks ** 2
        seq = torch.sqrt(torch.linspace(0.0, T ** 2, self.max_iter, device=self.device)).type(torch.int32)
        seq[-1] = seq[-1] - 1
        return rhos.to(self.device), sigmas.to(self.device), seq

    def find_nearest(self, array, value):
        return torch.abs(array - value).argmin()

    def get_alpha_prod(self, beta_start=0.1 / 1000, beta_end=20 / 1000, num_train_timesteps=1000):
        betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32, device=self.device)
        ac = torch.cumprod(1.0 - betas, dim=0)
        return torch.sqrt(1.0 / ac), torch.sqrt(1.0 / ac - 1)

    def _host_schedule(self):
        """Per-step scalars of the sample path, evaluated ONCE per call on the host from the device schedule (same fp32
        expressions as diffusion.py:452-507, element for element): the loop itself then launches no scalar kernels and
        never reads the device.  Returns one dict per step."""
        sig, seq = self.sigmas.detach().cpu().float(), self.seq.detach().cpu().long()
        rac = self.reduced_alpha_cumprod.detach().cpu().float()
        sac, s1m = self.sqrt_alphas_cumprod.detach().cpu().float(), self.sqrt_1m_alphas_cumprod.detach().cpu().float()
        rhos = self.rhos.detach().cpu().float()
        sqrt_recip = self.get_alpha_prod()[0].detach().cpu().float()
        nearest = lambda value: int(torch.abs(rac - value).argmin())
        sigma0 = float(self.sigma) if not isinstance(self.sigma, torch.Tensor) else float(self.sigma.detach().cpu())
        steps, n = [], len(seq)
        for i in range(n):
            curr_sigma = sig[seq[i]]
            t_i = nearest(curr_sigma)
            at = 1 / sqrt_recip[t_i] ** 2
            st = {"sigma_den": float(curr_sigma / 2), "pre_scale": float(1 / (2 * at.sqrt())), "last": int(seq[i]) == int(seq[-1])}
            if i == 0:
                st["init_noise"] = float((curr_sigma ** 2 - 4.0 * sigma0 ** 2).sqrt())
                st["init_div"] = float(sqrt_recip[-1])
            if not st["last"]:
                t_im1 = nearest(sig[seq[i + 1]])
                st["gamma"] = float(1.0 / (2 * rhos[t_i]))
                # x <- s_a[t-1] x0 + s_1ma[t-1] (sqrt(1 - zeta) eps + sqrt(zeta) n),  eps = (x - s_a[t] x0) / s_1ma[t]
                ca = float(s1m[t_im1]) * (1 - self.zeta) ** 0.5 / float(s1m[t_i])
                st["cx"], st["cx0"] = ca, float(sac[t_im1]) - ca * float(sac[t_i])
                st["cn"] = float(s1m[t_im1]) * self.zeta ** 0.5
                st["sa_t"], st["s1m_t"], st["sa_p"], st["s1m_p"] = float(sac[t_i]), float(s1m[t_i]), float(sac[t_im1]), float(s1m[t_im1])
            steps.append(st)
        return steps

    def forward(self, y, physics, seed=None, x_init=None):
        """diffusion.py:423-513.  The schedule look-ups happen once on the host (`_host_schedule`); on a HIP device each of the
        sampler's affine updates is ONE launch of csrc/elementwise.hip (`dinv_affine`):
            x_aux = x / (2 sqrt(a_t)) + 0.5                      -> affine(x)
            prox input clamp(2 D - 1, -1, 1) / 2 + 0.5 = clamp(D, 0, 1)   -> affine(D) with clamp
            x <- c_x x + c_x0 (2 x0_p - 1) + c_n n               -> affine(x, x0_p, n)
        (the Gaussian draws stay torch.randn_like: the reference's generator stream)."""
        from ..hip import elementwise as EW

        if seed:
            torch.manual_seed(seed)
        if hasattr(physics.noise_model, "sigma"):
            self.rhos, self.sigmas, self.seq = self.get_noise_schedule(sigma=physics.noise_model.sigma)
        steps = self._host_schedule()
        with torch.no_grad():
            x = physics.A_adjoint(y) if x_init is None else x_init
            fused = EW.eligible(x)
            x = EW.affine(2.0, x, d=-1.0) if fused else 2 * x - 1
            for i, st in enumerate(steps):
                if i == 0:
                    nz = torch.randn_like(x)
                    if fused and EW.eligible(nz):
                        x = EW.affine(1.0 / st["init_div"], x, st["init_noise"] / st["init_div"], nz)
                    else:
                        x = (x + st["init_noise"] * nz) / st["init_div"]
                x_aux = EW.affine(st["pre_scale"], x, d=0.5) if fused else x * st["pre_scale"] + 0.5
                den = self.model(x_aux, st["sigma_den"])
                if st["last"]:
                    continue
                dfused = fused and EW.eligible(den)
                z = EW.affine(1.0, den, lo=0.0, hi=1.0) if dfused else (2 * den - 1).clamp(-1, 1) / 2 + 0.5
                x0_p = self.data_fidelity.prox(z, y, physics, gamma=st["gamma"])
                nz = torch.randn_like(x)
                if dfused and EW.eligible(x0_p, nz):
                    x = EW.affine(st["cx"], x, 2.0 * st["cx0"], x0_p, st["cn"], nz, d=-st["cx0"])
                else:
                    x0 = x0_p * 2 - 1
                    eps = (x - st["sa_t"] * x0) / st["s1m_t"]
                    x = st["sa_p"] * x0 + st["s1m_p"] * (1 - self.zeta) ** 0.5 * eps + st["s1m_p"] * self.zeta ** 0.5 * nz
                    if fused:       # (a prox or model that returns another memory format: the fused updates read dense memory)
                        x = x.contiguous()
            return EW.affine(0.5, x, d=0.5) if fused else x / 2 + 0.5

    @property
    def _seq_host(self):
        # the step sequence is a host-side schedule; keep one CPU copy for control flow
        if getattr(self, "_seq_cpu_src", None) is not self.seq:
            self._seq_cpu_src = self.seq
            self._seq_cpu = self.seq.cpu()
        return self._seq_cpu
