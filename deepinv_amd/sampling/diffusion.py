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

    def forward(self, y, physics, seed=None, x_init=None):
        """diffusion.py:423-513: all schedule look-ups stay on the device (no .item() syncs)."""
        if seed:
            torch.manual_seed(seed)
        if hasattr(physics.noise_model, "sigma"):
            self.rhos, self.sigmas, self.seq = self.get_noise_schedule(sigma=physics.noise_model.sigma)
        x = 2 * (physics.A_adjoint(y) if x_init is None else x_init) - 1
        sqrt_recip, _ = self.get_alpha_prod()
        with torch.no_grad():
            n = len(self.seq)
            for i in range(n):
                curr_sigma = self.sigmas[self.seq[i]]
                t_i = self.find_nearest(self.reduced_alpha_cumprod, curr_sigma)
                at = 1 / sqrt_recip[t_i] ** 2
                if i == 0:
                    x = (x + (curr_sigma ** 2 - 4.0 * self.sigma ** 2).sqrt() * torch.randn_like(x)) / sqrt_recip[-1]
                x_aux = x / (2 * at.sqrt()) + 0.5
                denoised = 2 * self.model(x_aux, curr_sigma / 2) - 1
                x0 = denoised.clamp(-1, 1)
                if int(self._seq_host[i]) != int(self._seq_host[-1]):
                    x0_p = self.data_fidelity.prox(x0 / 2 + 0.5, y, physics, gamma=1.0 / (2 * self.rhos[t_i]))
                    x0 = x0_p * 2 - 1
                    t_im1 = self.find_nearest(self.reduced_alpha_cumprod, self.sigmas[self.seq[i + 1]])
                    eps = (x - self.sqrt_alphas_cumprod[t_i] * x0) / self.sqrt_1m_alphas_cumprod[t_i]
                    x = (self.sqrt_alphas_cumprod[t_im1] * x0
                         + self.sqrt_1m_alphas_cumprod[t_im1] * (1 - self.zeta) ** 0.5 * eps
                         + self.sqrt_1m_alphas_cumprod[t_im1] * self.zeta ** 0.5 * torch.randn_like(x))
        return x / 2 + 0.5

    @property
    def _seq_host(self):
        # the step sequence is a host-side schedule; keep one CPU copy for control flow
        if getattr(self, "_seq_cpu_src", None) is not self.seq:
            self._seq_cpu_src = self.seq
            self._seq_cpu = self.seq.cpu()
        return self._seq_cpu
