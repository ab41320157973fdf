#!/usr/bin/env python3
"""Workload for a rocprofv3 --pmc pass over the row-wise weight quantizers (exponent v1 / v2, learnable mixed-bit, uniform)
on Llama shapes, forward and backward, fp32 and fp16: every kernel is launched three times on a [4096, 4096] and once on
an [11008, 4096] and a [4096, 11008] matrix.
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \\
              SQ_INSTS_VALU SQ_WAVES --output-format csv -d <dir> -- python3 tools/row_pmc.py
tools/row_pmc_summary.py turns the counter file into the per-kernel table kept under profiles/."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "llm-quantization_amd"))
from miq import functional as F  # noqa: E402

DEV = "cuda:0"


def main():
    for dt in (torch.float32, torch.float16):
        for o, i, reps in ((4096, 4096, 3), (11008, 4096, 1), (4096, 11008, 1)):
            x = (torch.randn(o, i, device=DEV) * 0.02).to(dt)
            g = (torch.randn(o, i, device=DEV) * 1e-3).to(dt)
            up = torch.full((o, 1), 0.982, device=DEV, dtype=dt)
            low = torch.full((o, 1), 0.982, device=DEV, dtype=dt)
            for _ in range(reps):
                F.rowquant_uniform_fwd(x, 4, False, up, low)
                F.rowquant_uniform_bwd(g, x, 4, False, up, low, want_dx=True)
                F.rowquant_exp2_fwd(x, 3, up, low)
                F.rowquant_exp2_bwd(g, x, 3, up, low, want_dx=True)
                F.rowquant_exp1_fwd(x, 3, True, up, low)
                F.rowquant_exp1_bwd(g, x, 3, True, up, low, want_dx=True)
                if dt == torch.float32:
                    nb = 3.5 + torch.rand(o, 1, device=DEV) / 4
                    F.rowquant_mix_fwd(x, nb, False, up, low, hard=False)
                    F.rowquant_mix_bwd(g, x, nb, False, up, low, hard=False, want_dx=True)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
