"""RMSNorm at the LLaDA-8B width: one workgroup per row against one wave per row (llada_rmsnorm_use_wave_kernel):
bit-equality and time per call, input cold (a ring of buffers) and just written (L2 / Infinity Cache resident).
python tools/rmsnorm_bench.py [rows ...]"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load()
d = 4096
rows = [int(a) for a in sys.argv[1:]] or [384, 3072, 5120, 24576]
w = (1 + 0.1 * torch.randn(d, device="cuda")).to(torch.bfloat16)
for M in rows:
    NR = max(2, (600 << 20) // (M * d * 2))
    xs = [torch.randn(M, d, device="cuda").to(torch.bfloat16) for _ in range(NR)]
    out = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
    idx = torch.randperm(M, device="cuda")[: max(1, M // 12)].to(torch.int32)
    res = {}
    outs = {}
    for mode in (0, 1):
        lib.llada_rmsnorm_use_wave_kernel(mode)
        _lib.check(lib.llada_rmsnorm_bf16(xs[0].data_ptr(), w.data_ptr(), out.data_ptr(), M, d, 1e-5, None, None))
        o1 = out.clone()
        _lib.check(lib.llada_rmsnorm_bf16(xs[0].data_ptr(), w.data_ptr(), out.data_ptr(), idx.numel(), d, 1e-5, idx.data_ptr(), None))
        outs[mode] = (o1, out[: idx.numel()].clone())
        for name, ring in (("cold", NR), ("hot", 1)):
            ts = []
            for r in range(5):
                for i in range(ring): lib.llada_rmsnorm_bf16(xs[i % ring].data_ptr(), w.data_ptr(), out.data_ptr(), M, d, 1e-5, None, None)
                torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(40): lib.llada_rmsnorm_bf16(xs[i % ring].data_ptr(), w.data_ptr(), out.data_ptr(), M, d, 1e-5, None, None)
                e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 40 * 1e3)
            res[(mode, name)] = float(np.median(ts))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    print(f"rows {M:6d}: workgroup per row cold {res[(0,'cold')]:6.1f} hot {res[(0,'hot')]:6.1f} us | wave per row cold {res[(1,'cold')]:6.1f} hot {res[(1,'hot')]:6.1f} us"
          f" | {2 * M * d * 2 / res[(1,'cold')] * 1e-6:5.2f} TB/s cold | bit-identical={same}", flush=True)
lib.llada_rmsnorm_use_wave_kernel(1)
