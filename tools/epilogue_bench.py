"""HBM-bound row kernels at the full vocabulary: time per launch and achieved GB/s of the remasking
epilogue (all rules, with / without guidance and Gumbel noise) and of the masked cross-entropy.
Algorithmic bytes = one read of every logits row that is used (2 * V per row; twice under CFG).
Each timed launch works on a different buffer of a ring larger than the caches."""
import ctypes as C
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from llada_amd import _lib  # noqa: E402

lib = _lib.load()
V, MASK = 126464, 126336
rows_list = [int(a) for a in sys.argv[1:]] or [256, 32, 2048]


def timed(fn, n_ring, reps=3, per=12):
    ts = []
    for _ in range(reps):
        for i in range(2):
            fn(i % n_ring)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(per):
            fn(i % n_ring)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / per * 1e3)
    return float(np.median(ts))


out = []
for R in rows_list:
    bytes_row = 2 * V
    n_ring = max(2, min(12, int(1.2e9 // (R * bytes_row))))
    ring = [torch.randn(R, V, device="cuda").mul_(3).to(torch.bfloat16) for _ in range(n_ring)]
    un = torch.randn(R, V, device="cuda").mul_(3).to(torch.bfloat16)
    x = torch.full((1, R), MASK, dtype=torch.long, device="cuda")
    x0 = torch.empty_like(x)
    conf = torch.empty((1, R), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.llada_rows_workspace_bytes(R), dtype=torch.uint8, device="cuda")
    for use_ws in (True, False):
      for name, rule, temp, cfg in (("low_confidence T=0", "low_confidence", 0.0, 0.0), ("low_entropy T=0", "low_entropy", 0.0, 0.0),
                                  ("random T=0", "random", 0.0, 0.0), ("low_confidence T=0 cfg", "low_confidence", 0.0, 1.0),
                                  ("low_confidence T=0.7", "low_confidence", 0.7, 0.0)):
        if use_ws and "entropy" in rule:
            continue                    # the entropy rules always run the whole-row kernel
        def call(i):
            a = _lib.RemaskArgs(ring[i].data_ptr(), un.data_ptr() if cfg else None, cfg, None, R, V, x.data_ptr(), 1, R, R,
                                MASK, _lib.REMASK[rule], 0, 126081, 126348, temp, None, None, 1, 0, 0, x0.data_ptr(),
                                conf.data_ptr(), ws.data_ptr() if use_ws else None, ws.numel() if use_ws else 0)
            _lib.check(lib.llada_remask_scores(C.byref(a), st))
        t = timed(call, n_ring)
        nbytes = R * bytes_row * (2 if cfg else 1)
        rec = {"kernel": "remask_stream" if use_ws else "remask_rows", "case": name, "rows": R, "us": round(t, 1),
               "GB_s": round(nbytes / t * 1e-3, 1)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    seq = torch.randint(0, 126000, (1, R), device="cuda")
    pm = torch.full((1,), 0.5, device="cuda")
    rl = torch.empty(R, device="cuda")
    ls = torch.zeros(1, dtype=torch.float64, device="cuda")
    for use_ws, cfg in ((True, 0.0), (True, 1.0), (False, 0.0), (False, 1.0)):
        def call(i):
            _lib.check(lib.llada_masked_ce(ring[i].data_ptr(), un.data_ptr() if cfg else None, cfg, None, R, V, x.data_ptr(),
                                           seq.data_ptr(), 1, R, MASK, pm.data_ptr(), 1, rl.data_ptr(), ls.data_ptr(),
                                           ws.data_ptr() if use_ws else None, ws.numel() if use_ws else 0, st))
        t = timed(call, n_ring)
        nbytes = R * bytes_row * (2 if cfg else 1)
        rec = {"kernel": "masked_ce_stream" if use_ws else "masked_ce", "case": f"cfg={cfg}", "rows": R, "us": round(t, 1),
               "GB_s": round(nbytes / t * 1e-3, 1)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    del ring
