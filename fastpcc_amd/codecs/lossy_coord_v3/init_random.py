"""Seeded initialisation for tests and benchmarks: keeps activations O(1) over the levels and makes the latents take a
handful of distinct integer values (the default init rounds every latent to 0, which would test nothing)."""
import torch


def randomize_(model: torch.nn.Module, seed: int, gain: float = 1.0, latent_gain: float = 4.0) -> None:
    g = torch.Generator().manual_seed(seed)
    g_prior = torch.Generator().manual_seed(seed + 1000)
    u = lambda shape: torch.rand(shape, generator=g) * 2 - 1
    with torch.no_grad():
        for name, p in model.named_parameters():
            if '.prior_' in name:                              # deep-factorised prior keeps make_parameters' init; its
                if '.prior_biases.' in name:                   # biases (random there) come from their own seeded stream
                    p.copy_((torch.rand(p.shape, generator=g_prior) - 0.5).to(p.device))
                continue
            if name.endswith('.kernel'):
                k, c_in = (p.shape[0], p.shape[1]) if p.dim() == 3 else (1, p.shape[0])
                fan = c_in * {27: 13, 8: 4}.get(k, k)         # about half of the 27 neighbours exist on a surface
                w = u(p.shape) * gain * (3.0 / fan) ** 0.5
                if '.transforms.' in name and name.endswith('.1.4.kernel'):
                    w *= latent_gain
                p.copy_(w.to(p.device))
            elif name.endswith('.weight') and p.dim() == 2:     # nn.Linear [out, in]
                p.copy_((u(p.shape) * gain * (3.0 / p.shape[1]) ** 0.5).to(p.device))
            elif name.endswith('.weight') and p.dim() == 1:     # nn.PReLU slope
                p.copy_((0.1 + 0.3 * torch.rand(p.shape, generator=g)).to(p.device))
            else:                                               # biases
                p.copy_((u(p.shape) * 0.3).to(p.device))
