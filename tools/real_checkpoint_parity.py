#!/usr/bin/env python3
"""Real-weight parity of llada_amd against the reference stack, for a machine that has BOTH a local LLaDA
checkpoint directory (weights as *.safetensors, config.json, tokenizer files and the checkpoint's own
modeling_llada.py / configuration_llada.py) and the reference repository.  Nothing is downloaded.

    python tools/real_checkpoint_parity.py --checkpoint /path/to/LLaDA-8B-Base --reference /path/to/LLaDA

Both sides get the same token ids:
  reference side   AutoModel.from_pretrained(checkpoint, trust_remote_code=True, torch_dtype=bfloat16,
                   local_files_only=True) on the GPU (PyTorch-ROCm), driven by the reference's own
                   generate.py::generate (generate.py:43-136) and get_log_likelihood.py::get_log_likelihood (:46-77);
  this side        llada_amd.LLaDAModel(LLaDAWeights.from_checkpoint(checkpoint)) (safetensors loader: tensors only,
                   nothing from the checkpoint is executed) with llada_amd.generate / get_log_likelihood.

Reported, with the contract's bounds:
  1. per-step logits, max relative difference (contract: <= 2e-2 of the largest logit), on the reference
     demo's prompt shape (generate.py:139-175) for a fully masked answer and for a half-revealed one;
  2. top-k transfer: given the reference's own confidence tensor of a step, the transferred set (bit-exact;
     a tie at the k-th value admits any of the tied positions);
  3. final sequences at temperature 0 / low_confidence: fraction of identical tokens (contract: >= 99 %);
  4. log-likelihood (get_log_likelihood.py:80-92 style inputs, masks injected on both sides): relative
     difference (contract: <= 1e-3).
Exit status 0 when all four hold.  This script cannot run in the build container (no checkpoint, no network);
until it has been run on such a machine the forward's parity against the reference stays unpinned.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--checkpoint", required=True, help="local checkpoint directory (safetensors + modeling_llada.py)")
    ap.add_argument("--reference", required=True, help="checkout of the reference repository (generate.py, get_log_likelihood.py)")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--gen-length", type=int, default=128)
    ap.add_argument("--block-length", type=int, default=32)
    ap.add_argument("--mc-num", type=int, default=32)
    ap.add_argument("--batch-size", type=int, default=16)
    args = ap.parse_args()

    import numpy as np
    import torch

    sys.path.insert(0, os.path.abspath(args.reference))
    import generate as ref_generate                    # the reference's sampler
    import get_log_likelihood as ref_ll                # the reference's estimator
    from transformers import AutoModel, AutoTokenizer

    import llada_amd
    from llada_amd import LLaDAModel, LLaDAWeights

    device = torch.device(args.device)
    torch.cuda.set_device(device)
    import glob
    if not glob.glob(os.path.join(args.checkpoint, "*.safetensors")):
        sys.exit(f"{args.checkpoint} holds no *.safetensors files: both sides load tensors only (a pytorch_model*.bin "
                 "would be unpickled by transformers; convert the checkpoint first)")
    tok = AutoTokenizer.from_pretrained(args.checkpoint, trust_remote_code=True, local_files_only=True)
    ref_model = AutoModel.from_pretrained(args.checkpoint, trust_remote_code=True, torch_dtype=torch.bfloat16,
                                          local_files_only=True, use_safetensors=True).to(device).eval()
    model = LLaDAModel(LLaDAWeights.from_checkpoint(args.checkpoint, device))
    mask_id = model.cfg.mask_token_id
    G, bl = args.gen_length, args.block_length

    # the reference demo's prompts (generate.py:150-160): three questions, left-padded, with attention mask
    questions = ["Lily can run 12 kilometers per hour for 4 hours. After that, she runs 6 kilometers per hour. "
                 "How many kilometers can she run in 8 hours?",
                 "Joy can read 8 pages of a book in 20 minutes. How many hours will it take her to read 120 pages?",
                 "Randy has 60 mango trees on his farm. He also has 5 less than half as many coconut trees as mango trees. "
                 "How many trees does Randy have in all on his farm?"]
    tok.padding_side = "left"
    enc = tok(questions, add_special_tokens=False, padding=True, return_tensors="pt")
    prompt, am = enc["input_ids"].to(device), enc["attention_mask"].to(device)
    B, P = prompt.shape
    report = {}

    # ---- 1. logits -------------------------------------------------------------------------------------
    worst = 0.0
    with torch.no_grad():
        base = torch.full((B, P + G), mask_id, dtype=torch.long, device=device)
        base[:, :P] = prompt
        am_full = torch.cat([am, torch.ones((B, G), dtype=am.dtype, device=device)], dim=-1)
        half = ref_generate.generate(ref_model, prompt, am, steps=G // 2, gen_length=G, block_length=G, temperature=0.,
                                     cfg_scale=0., remasking="low_confidence", mask_id=mask_id)
        half = torch.where(torch.rand(half.shape, device=device) < 0.5, half, base)     # a half-revealed state
        half[:, :P] = prompt
        for x in (base, half):
            want = ref_model(x, attention_mask=am_full).logits.float()
            got = model(x, attention_mask=am_full).logits.float()
            real = am_full.bool()
            rel = ((got - want).abs()[real].max() / want.abs()[real].max()).item()
            worst = max(worst, rel)
    report["logits_max_rel"] = worst
    report["logits_ok"] = worst <= 2e-2

    # ---- 2. top-k transfer on the reference's own confidence tensors -------------------------------------
    recorded = []
    real_topk = torch.topk

    def spy(conf_row, k, *a, **kw):
        out = real_topk(conf_row, k, *a, **kw)
        recorded.append((conf_row.detach().float().clone(), int(k), out.indices.detach().clone()))
        return out
    torch.topk = spy
    try:
        ref_final = ref_generate.generate(ref_model, prompt, am, steps=G, gen_length=G, block_length=bl, temperature=0.,
                                          cfg_scale=0., remasking="low_confidence", mask_id=mask_id)
    finally:
        torch.topk = real_topk
    lib = llada_amd._lib.load()
    exact = ties = 0
    for conf_row, k, idx in recorded:
        L = conf_row.numel()
        x = torch.zeros(1, L, dtype=torch.long, device=device)
        x0 = torch.ones(1, L, dtype=torch.long, device=device)
        sched = torch.tensor([[k]], dtype=torch.int32, device=device)
        c = conf_row.reshape(1, L).contiguous()
        llada_amd._lib.check(lib.llada_topk_transfer(x.data_ptr(), x0.data_ptr(), c.data_ptr(), 1, L, sched.data_ptr(), 1, 0,
                                                     llada_amd._lib.stream_ptr()))
        mine = set(torch.nonzero(x[0]).reshape(-1).tolist())
        theirs = set(idx.tolist())
        if mine == theirs:
            exact += 1
        else:
            kth = torch.sort(conf_row, descending=True).values[k - 1]
            sym = mine ^ theirs
            if all(conf_row[i] == kth for i in sym):
                ties += 1
    report["topk_steps"] = len(recorded)
    report["topk_exact"] = exact
    report["topk_differ_only_by_a_boundary_tie"] = ties
    report["topk_ok"] = exact + ties == len(recorded)

    # ---- 3. final sequences --------------------------------------------------------------------------------
    mine = llada_amd.generate(model, prompt, am, steps=G, gen_length=G, block_length=bl, temperature=0., cfg_scale=0.,
                              remasking="low_confidence", mask_id=mask_id)
    same = (mine[:, P:] == ref_final[:, P:]).float().mean().item()
    report["final_token_identity"] = same
    report["final_ok"] = same >= 0.99
    report["sample_reference"] = tok.batch_decode(ref_final[:1, P:], skip_special_tokens=True)[0][:200]
    report["sample_llada_amd"] = tok.batch_decode(mine[:1, P:], skip_special_tokens=True)[0][:200]

    # ---- 4. log-likelihood (get_log_likelihood.py:80-92) ---------------------------------------------------
    p_ids = torch.tensor(tok("Roof shingle removal: A man is sitting on a roof. He")["input_ids"], device=device)
    a_ids = torch.tensor(tok(" is using wrap to wrap a pair of skis.")["input_ids"], device=device)
    T = len(a_ids)
    n_iter = args.mc_num // args.batch_size
    g = np.random.Generator(np.random.PCG64(7))
    ks = [int(g.integers(1, T + 1)) for _ in range(n_iter)]
    perms = [g.permutation(T) for _ in range(n_iter * args.batch_size)]
    # the reference draws k with torch.randint and the per-row subsets with torch.randperm: inject the same draws
    k_it, p_it = iter(ks), iter(perms)
    real_randint, real_randperm = torch.randint, torch.randperm
    torch.randint = lambda lo, hi, size, **kw: torch.full(size, next(k_it), dtype=torch.long, device=kw.get("device"))
    torch.randperm = lambda n, **kw: torch.from_numpy(next(p_it).copy())
    try:
        want = ref_ll.get_log_likelihood(ref_model, p_ids, a_ids, mc_num=args.mc_num, batch_size=args.batch_size, mask_id=mask_id)
    finally:
        torch.randint, torch.randperm = real_randint, real_randperm
    k2, p2 = iter(ks), iter(perms)
    got = llada_amd.get_log_likelihood(model, p_ids, a_ids, mc_num=args.mc_num, batch_size=args.batch_size, mask_id=mask_id,
                                       k_fn=lambda T_: next(k2), perm_fn=lambda T_: next(p2))
    report["ll_reference"], report["ll_llada_amd"] = want, got
    report["ll_rel"] = abs(got - want) / abs(want)
    report["ll_ok"] = report["ll_rel"] <= 1e-3

    print(json.dumps(report, indent=1))
    ok = report["logits_ok"] and report["topk_ok"] and report["final_ok"] and report["ll_ok"]
    print("PARITY OK" if ok else "PARITY FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
