"""How far the logits of one row move when only the GEMM tiling changes (the row alone: split-K small-M path;
the same row 8 times in a batch: full tiles), against the number of layers, LLaDA-8B widths, random weights."""
import sys

import torch

sys.path.insert(0, ".")
from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights  # noqa: E402

g = torch.Generator().manual_seed(5)
l = 400
row = torch.randint(0, 126000, (1, l), generator=g).cuda()
rows_idx = torch.arange(l - 8, l, dtype=torch.int32).cuda()
for n_layers in (1, 2, 4, 8, 16, 32):
    model = LLaDAModel(LLaDAWeights.random(LLaDAConfig(n_layers=n_layers), "cuda"))
    a = model.forward_rows(row, None, rows_idx).float()
    c = model.forward_rows(row.repeat(8, 1), None, rows_idx).float()
    d = (c - a).abs()
    print(f"{n_layers:2d} layers: max |diff| / max |logit| = {float(d.max() / a.abs().max()):.2e}, mean |diff| / max |logit| = "
          f"{float(d.mean() / a.abs().max()):.2e}, entries that differ {float((d > 0).float().mean()):.3f}", flush=True)
    del model
    torch.cuda.empty_cache()
