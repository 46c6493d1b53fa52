"""Run the attention kernel a few times (for rocprofv3 passes): attn_one.py [n_seq] [L] [launches]."""
import sys, torch
sys.path.insert(0, ".")
from llada_amd import _lib
lib = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 384
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
H = 32; d = H * 128
qkv = torch.randn(B * L, 3 * d, device="cuda").to(torch.bfloat16)
out = torch.zeros(B * L, d, device="cuda", dtype=torch.bfloat16)
for _ in range(n): lib.llada_attn_bidir_bf16(qkv.data_ptr(), None, out.data_ptr(), B, L, H, None)
torch.cuda.synchronize()
