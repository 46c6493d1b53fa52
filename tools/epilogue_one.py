"""A few launches of the temperature-0 low_confidence epilogue at the full vocabulary, for
`rocprofv3 --kernel-trace --stats -- python3 tools/epilogue_one.py ROWS [whole]`."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from llada_amd import _lib  # noqa: E402

lib = _lib.load()
V, MASK = 126464, 126336
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
whole = len(sys.argv) > 2 and sys.argv[2] == "whole"
ring = [torch.randn(R, V, device="cuda").mul_(3).to(torch.bfloat16) for _ in range(8)]
x = torch.full((1, R), MASK, dtype=torch.long, device="cuda")
x0 = torch.empty_like(x)
conf = torch.empty((1, R), dtype=torch.float32, device="cuda")
ws = torch.empty(lib.llada_rows_workspace_bytes(R), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for i in range(40):
    a = _lib.RemaskArgs(ring[i % 8].data_ptr(), None, 0.0, None, R, V, x.data_ptr(), 1, R, R, MASK,
                        _lib.REMASK["low_confidence"], 0, 126081, 126348, 0.0, None, None, 1, 0, 0, x0.data_ptr(),
                        conf.data_ptr(), None if whole else ws.data_ptr(), 0 if whole else ws.numel())
    _lib.check(lib.llada_remask_scores(C.byref(a), st))
torch.cuda.synchronize()
