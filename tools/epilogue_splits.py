"""Sweep of the pieces-per-row of the streamed epilogue (test hook llada_rows_force_splits): time per
llada_remask_scores call, temperature 0, low_confidence, full vocabulary, buffers colder than the caches."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from llada_amd import _lib  # noqa: E402

lib = _lib.load()
V, MASK = 126464, 126336
st = torch.cuda.current_stream().cuda_stream
for R in [int(a) for a in sys.argv[1:]] or [256, 32, 128, 512, 1024]:
    n_ring = max(2, min(12, int(1.2e9 // (R * 2 * V))))
    ring = [torch.randn(R, V, device="cuda").mul_(3).to(torch.bfloat16) for _ in range(n_ring)]
    x = torch.full((1, R), MASK, dtype=torch.long, device="cuda")
    x0 = torch.empty_like(x)
    conf = torch.empty((1, R), dtype=torch.float32, device="cuda")
    ws = torch.empty(lib.llada_rows_workspace_bytes(R), dtype=torch.uint8, device="cuda")
    for S in (1, 2, 4, 8):
        lib.llada_rows_force_splits(S)
        ts = []
        for rep in range(3):
            def call(i):
                a = _lib.RemaskArgs(ring[i % n_ring].data_ptr(), None, 0.0, None, R, V, x.data_ptr(), 1, R, R, MASK,
                                    _lib.REMASK["low_confidence"], 0, 126081, 126348, 0.0, None, None, 1, 0, 0,
                                    x0.data_ptr(), conf.data_ptr(), ws.data_ptr(), ws.numel())
                _lib.check(lib.llada_remask_scores(C.byref(a), st))
            for i in range(3):
                call(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(16):
                call(i)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 16 * 1e3)
        t = float(np.median(ts))
        print(f"rows {R:5d} splits {S}: {t:7.1f} us  {R * 2 * V / t * 1e-3:7.1f} GB/s", flush=True)
    lib.llada_rows_force_splits(0)
    del ring
