"""Deterministic synthetic inputs shared by the golden-vector generator, the tests and bench.py.

Everything is generated on the CPU with an explicit ``torch.Generator`` so the same (recipe, seed,
shape) gives the same bytes in the build container and on the GPU box (same torch build).  Tensors
are moved to the GPU afterwards by the caller.

The recipes follow the shapes the calibration loop sees: weights are ``randn*0.02`` with a few
outlier input channels, activations are ``randn`` with a per-channel spread over two decades so that
the range-sorted channel reorder is non-trivial, LWC factors are ``4 + 0.5*randn`` (the reference
initialises them to 4.0, quantize/quantizer.py:62-73, and the optimiser moves them a little).
"""
from __future__ import annotations

import hashlib

import torch


def gen(seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))
    return g


def weight(rows: int, cols: int, seed: int = 2, dtype=torch.float32, outlier_frac: float = 0.001):
    g = gen(seed)
    w = torch.randn(rows, cols, generator=g) * 0.02
    n_out = max(1, int(cols * outlier_frac))
    which = torch.randperm(cols, generator=g)[:n_out]
    w[:, which] *= 20.0
    return w.to(dtype)


def weight_with_edge_rows(rows: int, cols: int, seed: int = 2, dtype=torch.float32):
    """``weight`` plus hand-made rows that exercise the corner cases of the row-wise quantizers.

    row 0: constant non-zero row      row 1: all-zero row        row 2: strictly negative row
    row 3: strictly positive row      row 4: contains +0 and -0   row 5: maximum attained twice
    row 6: minimum attained 3 times   row 7: exact .5 rounding ties on a power-of-two grid
    row 8: values on exp-quantizer thresholds (1.5 * 2^k of the row maximum / 2qmax)
    """
    assert rows >= 10 and cols >= 16
    w = weight(rows, cols, seed, torch.float32)
    g = gen(seed + 1000)
    w[0] = 0.03125
    w[1] = 0.0
    w[2] = -(torch.rand(cols, generator=g) + 0.25)
    w[3] = torch.rand(cols, generator=g) + 0.25
    w[4, 0] = 0.0
    w[4, 1] = -0.0
    w[5, 3] = w[5].max() + 0.125
    w[5, 11] = w[5, 3]
    m = w[6].min() - 0.25
    w[6, 2] = m
    w[6, 7] = m
    w[6, 13] = m
    # row 7: range exactly [0, 7] + ties: with 3 bits (Q=7) and no clipping scale is exactly 1
    w[7] = torch.randint(0, 8, (cols,), generator=g).float()
    w[7, 0] = 0.0
    w[7, 1] = 7.0
    w[7, 2] = 0.5
    w[7, 3] = 1.5
    w[7, 4] = 2.5
    w[7, 5] = 3.5
    w[7, 6] = 6.5
    # row 8: absmax 8 -> exp-v2 3-bit scale = 8/(2*4) = 1 -> |q| hits 0.5, 0.75, 1.5, 3, 6 exactly
    w[8] = (torch.rand(cols, generator=g) - 0.5) * 8.0
    w[8, 0] = 8.0
    w[8, 1] = 0.5
    w[8, 2] = -0.5
    w[8, 3] = 0.75
    w[8, 4] = 1.5
    w[8, 5] = -1.5
    w[8, 6] = 3.0
    w[8, 7] = 6.0
    w[8, 8] = 0.49999997
    w[8, 9] = 2.9999998
    return w.to(dtype)


def lwc_factors(n: int, seed: int = 3, dtype=torch.float32):
    g = gen(seed)
    up = (4.0 + 0.5 * torch.randn(n, 1, generator=g)).to(dtype)
    low = (4.0 + 0.5 * torch.randn(n, 1, generator=g)).to(dtype)
    return up, low


def lwc_clips(n: int, seed: int = 3, dtype=torch.float32):
    """The activated clip factors (sigmoid of lwc_factors): what the row-wise kernels take."""
    up, low = lwc_factors(n, seed, dtype)
    return torch.sigmoid(up), torch.sigmoid(low)


def activation(shape, seed: int = 2, dtype=torch.float32):
    """randn with per-channel std logspace(-1, 1, C) (shuffled) and per-channel mean offsets."""
    g = gen(seed)
    c = shape[-1]
    std = torch.logspace(-1, 1, c)[torch.randperm(c, generator=g)]
    mean = torch.randn(c, generator=g) * 0.5
    x = torch.randn(*shape, generator=g) * std + mean
    return x.to(dtype)


def grad_like(t: torch.Tensor, seed: int = 7):
    g = gen(seed)
    return torch.randn(t.shape, generator=g).to(t.dtype)


def mix_bits(rows: int, target_bit: float = 3.0, seed: int = 5):
    """Initial learnable bit-widths: target + 0.5 + rand/4 (quantize/quantizer_mix.py:63)."""
    g = gen(seed)
    return torch.ones(rows, 1) * target_bit + 0.5 + torch.rand(rows, 1, generator=g) / 4


def mix2_bits(rows: int, seed: int = 5):
    """Fixed per-row bits: 10% 4-bit, 80% 3-bit, 10% 2-bit in shuffled order (get_grads.py:223-231)."""
    g = gen(seed)
    n4 = rows // 10
    n2 = rows // 10
    bits = torch.full((rows,), 3.0)
    perm = torch.randperm(rows, generator=g)
    bits[perm[:n4]] = 4.0
    bits[perm[n4:n4 + n2]] = 2.0
    return bits


def permutation(n: int, seed: int = 11):
    return torch.randperm(n, generator=gen(seed))


def sha(t: torch.Tensor) -> str:
    """Content hash of a tensor's bytes (dtype- and shape-sensitive)."""
    t = t.detach().cpu().contiguous()
    h = hashlib.sha256()
    h.update(str(t.dtype).encode())
    h.update(str(tuple(t.shape)).encode())
    h.update(t.view(torch.uint8).numpy().tobytes() if t.numel() else b"")
    return h.hexdigest()[:32]
