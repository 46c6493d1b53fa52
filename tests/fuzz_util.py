"""Randomised end-to-end check of llada_amd.generate on a small LLaDA-shaped HIP model (GPU only; the oracle
is the checker): random batch / prompt / generation / block / step shapes, left padding, all five remasking
rules, temperature 0 and 0.7 (injected noise), classifier-free guidance.  Every step is teacher-forced: the
oracle's epilogue on the HIP model's own logits of that step's input (requested exactly as generate() requests
them) must give the same candidates and confidences, the transfers must be the oracle's top-k of them under
the schedule of generate.py:75-79, and the forward's logits must stay within 2e-2 (max-relative) of the
oracle model's.  Used by tests/test_fuzz_gpu.py (fixed seeds) and tools/fuzz_generate.py (any seed)."""
import numpy as np
import torch

from oracle import sampler as osamp
from oracle.bf16 import ulp_distance
from oracle.model import LLaDAConfig as OCfg
from oracle.model import forward, init_weights, positions_from_mask

V, MASK_ID = 512, 511      # the mask token is an ordinary word of this small vocabulary
RULES = ["low_confidence", "high_confidence", "low_entropy", "high_entropy", "random"]


def build_model():
    from llada_amd import LLaDAConfig, LLaDAModel, LLaDAWeights
    ocfg = OCfg(d_model=256, n_layers=2, n_heads=2, mlp_hidden=512, vocab_size=V)
    W = init_weights(ocfg, seed=3, std=0.05, head_scale=4.0)
    cfg = LLaDAConfig(d_model=256, n_layers=2, n_heads=2, mlp_hidden=512, vocab_size=V)
    return ocfg, W, LLaDAModel(LLaDAWeights.from_numpy(cfg, W, "cuda"))


def run(n_iter, seed, verbose=False, built=None, cache=None):
    """Runs n_iter random cases; raises AssertionError (with the case description) on the first mismatch.
    Returns the worst forward max-relative error seen.  ``cache="dual"``: the block-cached sampler (an approximation
    of its own; its steps are replayed on a second cache with the calls generate() makes, so the epilogue and the
    transfers are checked on bit-identical logits; only the block's positions are candidates there); ``"prefix"``: the
    same with the block and everything right of it recomputed.  Padded cases of the uncached sampler number the
    rotary positions from every row's first real token half of the time (generate(position_ids="from_mask"))."""
    from llada_amd import generate
    ocfg, W, model = built if built is not None else build_model()
    g = np.random.Generator(np.random.PCG64(seed))
    worst_rel = 0.0
    for it in range(n_iter):
        B = int(g.integers(1, 6))
        P = int(g.integers(1, 70))
        bl = int(g.choice([1, 2, 4, 8, 16, 32, 48]))
        nb = int(g.integers(1, 4))
        G = bl * nb
        spb = int(g.integers(1, bl + 1))             # steps per block
        steps = spb * nb
        rule = RULES[int(g.integers(0, 5))]
        T = float(g.choice([0.0, 0.7]))
        cfg_scale = float(g.choice([0.0, 0.0, 1.5]))
        padded = bool(g.integers(0, 2)) or cfg_scale > 0
        prompt = g.integers(0, MASK_ID, size=(B, P), dtype=np.int64)
        am = None
        if padded:
            am = np.ones((B, P), dtype=np.int64)
            for b in range(B):
                am[b, : int(g.integers(0, max(1, P // 2)))] = 0
        L = P + G
        own_positions = padded and cache is None and bool(g.integers(0, 2))
        noise_log, rand_log, trace = [], [], []

        def noise_fn(shape):
            u = g.random(shape)
            noise_log.append(u)
            return torch.from_numpy(u)

        def rand_fn(shape):
            r = g.random(shape).astype(np.float32)
            rand_log.append(r)
            return torch.from_numpy(r)

        def cb(block, step, x, x0, conf):
            trace.append((block, x.cpu().numpy().copy(), x0.cpu().numpy().copy(), conf.cpu().numpy().copy()))

        desc = (f"it {it}: B={B} P={P} G={G} bl={bl} steps={steps} rule={rule} T={T} cfg={cfg_scale} padded={padded} "
                f"own_positions={own_positions}")
        out = generate(model, torch.from_numpy(prompt), None if am is None else torch.from_numpy(am), steps=steps, gen_length=G,
                       block_length=bl, temperature=T, cfg_scale=cfg_scale, remasking=rule, mask_id=MASK_ID,
                       noise_fn=noise_fn if T else None, rand_fn=rand_fn if rule == "random" else None,
                       step_callback=cb, cache=cache, position_ids="from_mask" if own_positions else None).cpu().numpy()
        kv2 = model.kv_cache(2 * B if cfg_scale > 0 else B, L) if cache else None
        assert np.array_equal(out[:, :P], prompt), desc
        # (the mask token is an ordinary word of this small vocabulary: the model may predict it, and such a
        # position then stays masked, in the reference as here, so "no mask token left" is not asserted)
        full_mask = None if am is None else np.concatenate([am, np.ones((B, G), np.int64)], 1)
        opos = positions_from_mask(full_mask) if own_positions else None           # the oracle's numbering
        pos1 = None if opos is None else torch.from_numpy(opos.astype(np.int32)).cuda()
        pos2 = None if opos is None else torch.cat([pos1, pos1])
        x_in = np.full((B, L), MASK_ID, dtype=np.int64)
        x_in[:, :P] = prompt
        prompt_index = x_in != MASK_ID
        for t, (block, x_after, x0, conf) in enumerate(trace):
            # the step's logits exactly as generate() asks for them: the rows that can matter (the block, or everything
            # left of its end when an earlier position still holds the mask token), cond | uncond halves under guidance
            bs_, be_ = P + block * bl, P + (block + 1) * bl
            if t % spb == 0:
                row_lo = 0 if (cache is None and bs_ > 0 and (x_in[:, :bs_] == MASK_ID).any()) else bs_
            rows = (np.arange(B)[:, None] * L + np.arange(row_lo, be_)[None, :]).reshape(-1).astype(np.int32)
            fm = None if full_mask is None else torch.from_numpy(full_mask).cuda()
            ce_ = be_ if cache == "dual" else L          # cached steps recompute [bs_, ce_); logits for the block's rows

            def cached_rows(n_seq):
                if cache != "prefix":
                    return None
                return (torch.arange(n_seq, dtype=torch.int32)[:, None] * (ce_ - bs_)
                        + torch.arange(be_ - bs_, dtype=torch.int32)[None, :]).reshape(-1).cuda().contiguous()
            lg = np.zeros((B, L, V), dtype=np.float32)
            if cfg_scale > 0:
                un = np.where(prompt_index, MASK_ID, x_in)
                both = np.concatenate([x_in, un])
                fm2 = torch.from_numpy(np.concatenate([full_mask, full_mask])).cuda()
                rows2 = torch.from_numpy(np.concatenate([rows, rows + B * L])).cuda()
                if cache and t % spb:
                    got = model.forward_block(torch.from_numpy(np.ascontiguousarray(both[:, bs_:ce_])).cuda(), fm2, L, bs_, kv2,
                                              update_cache=False, logit_rows=cached_rows(2 * B)).float().cpu().numpy()
                else:
                    got = model.forward_rows(torch.from_numpy(both).cuda(), fm2, rows2, kv_cache=kv2,
                                             position_ids=pos2).float().cpu().numpy()
                lg.reshape(B * L, V)[rows] = osamp.cfg_combine(got[: rows.size], got[rows.size:], cfg_scale)
            else:
                if cache and t % spb:
                    got = model.forward_block(torch.from_numpy(np.ascontiguousarray(x_in[:, bs_:ce_])).cuda(), fm, L, bs_, kv2,
                                              update_cache=False, logit_rows=cached_rows(B)).float().cpu().numpy()
                else:
                    got = model.forward_rows(torch.from_numpy(x_in).cuda(), fm, torch.from_numpy(rows).cuda(),
                                             kv_cache=kv2, position_ids=pos1).float().cpu().numpy()
                lg.reshape(B * L, V)[rows] = got
            if t == 0 and cfg_scale == 0:
                ref = forward(W, ocfg, x_in, full_mask, emulate_bf16=True, position_ids=opos).reshape(B * L, V)[rows]
                rel = float(np.abs(got - ref).max() / np.abs(ref).max())
                worst_rel = max(worst_rel, rel)
                assert rel <= 2e-2, (desc, rel)
            u = noise_log[t] if T else None
            rs = rand_log[t] if rule == "random" else None
            ox0, oconf = osamp.confidence_step(x_in, lg, MASK_ID, P + (block + 1) * bl, rule, T, u, rs)
            if cache:
                oconf[:, :bs_] = -np.inf
                ox0[:, :bs_] = x_in[:, :bs_]
            live = ~np.isneginf(oconf)
            assert np.array_equal(np.isneginf(conf), np.isneginf(oconf)), (desc, t)
            assert np.array_equal(x0[live], ox0[live]), (desc, t)
            fin = live & np.isfinite(oconf)
            if rule == "random":
                assert np.array_equal(conf[fin], oconf[fin]), (desc, t)
            else:
                # 1 - p inherits one bf16 step of p (2^-8 for p in [0.5, 1)) whatever its own magnitude
                tol = 2.0 ** -8 if rule == "high_confidence" else 2e-3
                bad = (ulp_distance(conf[fin], oconf[fin]) > 1) & (np.abs(conf[fin] - oconf[fin]) > tol)
                assert not bad.any(), (desc, t, conf[fin][bad][:4], oconf[fin][bad][:4])
            if t % spb == 0:      # generate.py:75-79: the schedule is drawn once per block
                sched = osamp.get_num_transfer_tokens(x_in[:, P + block * bl: P + (block + 1) * bl] == MASK_ID, spb)
            expect = x_in.copy()
            osamp.topk_transfer(expect, x0, conf, sched[:, t % spb])      # lowest index wins a tie, as in the library
            assert np.array_equal(x_after, expect), (desc, t)
            x_in = x_after
        assert np.array_equal(x_in, out), desc
        if verbose:
            print("ok", desc, flush=True)
    return worst_rel


def run_likelihood(n_iter, seed, built=None):
    """Random get_log_likelihood cases (prompt / answer lengths, batch size, iteration count, guidance, bf16 or
    float32 cross-entropies, injected k and permutations) on the HIP model against the oracle's estimator fed
    with the HIP model's own full logits; returns the worst relative difference (contract: 1e-3)."""
    from llada_amd import get_log_likelihood
    from oracle import likelihood as oll
    ocfg, W, model = built if built is not None else build_model()
    g = np.random.Generator(np.random.PCG64(seed))

    def hip_logits(ids, attention_mask=None):
        return model(torch.from_numpy(np.ascontiguousarray(ids)).cuda()).logits.float().cpu().numpy()

    worst = 0.0
    for it in range(n_iter):
        l1, T = int(g.integers(0, 40)), int(g.integers(1, 70))
        bs = int(g.choice([1, 2, 3, 4, 8, 16]))
        n_it = int(g.integers(1, 4))
        cfg_scale = float(g.choice([0.0, 0.0, 0.5]))
        rounded = bool(g.integers(0, 2))
        prompt = g.integers(0, MASK_ID, size=l1, dtype=np.int64)
        answer = g.integers(0, MASK_ID, size=T, dtype=np.int64)
        ks = [int(g.integers(1, T + 1)) for _ in range(n_it)]
        perms = [g.permutation(T) for _ in range(n_it * bs)]
        desc = f"it {it}: l1={l1} T={T} bs={bs} iters={n_it} cfg={cfg_scale} bf16_ce={rounded}"
        k1, p1 = list(ks), iter(perms)
        val = get_log_likelihood(model, torch.from_numpy(prompt), torch.from_numpy(answer), mc_num=n_it * bs, batch_size=bs,
                                 cfg_scale=cfg_scale, mask_id=MASK_ID, k_fn=lambda t: k1.pop(0), perm_fn=lambda t: next(p1),
                                 round_ce_bf16=rounded)
        k2, p2 = list(ks), iter(perms)
        ref = oll.get_log_likelihood(hip_logits, prompt, answer, mc_num=n_it * bs, batch_size=bs, cfg_scale=cfg_scale,
                                     mask_id=MASK_ID, k_fn=lambda t: k2.pop(0), perm_fn=lambda t: next(p2),
                                     round_to_bf16=rounded)
        rel = abs(val - ref) / abs(ref)
        worst = max(worst, rel)
        assert rel <= 1e-3, (desc, val, ref)
    return worst


def run_attention(n_iter, seed):
    """Random (sequences, length, heads, key-padding, waves per workgroup, query subset + key splits) calls of
    the attention entry points against the oracle's float32 attention; returns the worst error relative to the
    largest reference value (bound used by the kernel tests: 1.5e-2)."""
    import ctypes as C

    from llada_amd import _lib
    from oracle import model as omodel
    from oracle.bf16 import round_bf16
    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(seed))
    st = torch.cuda.current_stream().cuda_stream
    worst = 0.0
    for it in range(n_iter):
        n_seq, L, H = int(g.integers(1, 5)), int(g.integers(1, 700)), int(g.integers(1, 4))
        d = H * 128
        waves = int(g.choice([0, 0, 1, 2, 4]))
        qkv = round_bf16(g.standard_normal((n_seq * L, 3 * d), dtype=np.float32))
        qkv[g.integers(0, n_seq * L), d:2 * d] *= 5            # a late jump of the running maximum
        km = None
        if g.integers(0, 2):
            km = (g.random((n_seq, L)) > 0.3).astype(np.uint8)
            km[:, int(g.integers(0, L))] = 1                      # at least one key per sequence
        qt = torch.from_numpy(qkv).cuda().to(torch.bfloat16)
        kmt = None if km is None else torch.from_numpy(km).cuda()
        x = qkv.reshape(n_seq, L, 3, H, 128)
        q, k, v = (x[:, :, i].transpose(0, 2, 1, 3) for i in range(3))
        ref = omodel.attention(q, k, v, km, False).transpose(0, 2, 1, 3).reshape(n_seq, L, d)
        desc = f"it {it}: n_seq={n_seq} L={L} H={H} waves={waves} masked={km is not None}"
        lib.llada_attn_force_variant(waves)
        try:
            out = torch.full((n_seq * L, d), float("nan"), dtype=torch.bfloat16, device="cuda")
            _lib.check(lib.llada_attn_bidir_bf16(qt.data_ptr(), _lib.ptr(kmt), out.data_ptr(), n_seq, L, H, st))
            torch.cuda.synchronize()
            got = out.float().cpu().numpy().reshape(n_seq, L, d)
            assert np.isfinite(got).all(), desc
            err = float(np.abs(got - ref).max() / np.abs(ref).max())
            worst = max(worst, err)
            assert err <= 1.5e-2, (desc, err)
            # a query subset [q_lo, q_lo + n_q) of every sequence, with and without the key-split workspace
            n_q = int(g.integers(1, min(L, 96) + 1))
            q_lo = int(g.integers(0, L - n_q + 1))
            qs = qt.reshape(n_seq, L, 3 * d)[:, q_lo:q_lo + n_q].reshape(n_seq * n_q, 3 * d).contiguous()
            for use_ws in (False, True):
                nbytes = lib.llada_attn_workspace_bytes(n_seq, n_q, L, H) if use_ws else 0
                ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda") if use_ws and nbytes else None
                o2 = torch.full((n_seq * n_q, d), float("nan"), dtype=torch.bfloat16, device="cuda")
                _lib.check(lib.llada_attn_bidir_queries_bf16(qs.data_ptr(), 3 * d, n_q, qt.data_ptr(), _lib.ptr(kmt),
                                                             o2.data_ptr(), n_seq, L, H, None, 0, _lib.ptr(ws),
                                                             0 if ws is None else ws.numel(), st))
                torch.cuda.synchronize()
                g2 = o2.float().cpu().numpy().reshape(n_seq, n_q, d)
                e2 = float(np.abs(g2 - ref[:, q_lo:q_lo + n_q]).max() / np.abs(ref).max())
                worst = max(worst, e2)
                assert np.isfinite(g2).all() and e2 <= 1.5e-2, (desc, n_q, q_lo, use_ws, e2)
        finally:
            lib.llada_attn_force_variant(0)
    return worst


def run_gemm(n_iter, seed):
    """Random projection shapes (M any, N % 8 == 0, K % 64 == 0), plain / residual epilogue, forced tile height,
    split-K workspace, tile-major weights, on small-integer operands: every partial sum is exact in float32, so
    the result must EQUAL bf16(A W^T [+ R]) whatever the tiling and summation order."""
    from llada_amd import _lib
    from llada_amd.weights import tile_major
    from oracle.bf16 import round_bf16
    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(seed))
    st = torch.cuda.current_stream().cuda_stream
    try:
        for it in range(n_iter):
            M = int(g.integers(1, 900))
            K = 64 * int(g.integers(1, 25))
            tiled = bool(g.integers(0, 2))
            N = 256 * int(g.integers(1, 6)) if tiled else 8 * int(g.integers(1, 161))
            epi = int(g.integers(0, 2))                      # PLAIN / RESIDUAL
            rows = int(g.choice([0, 128, 192, 256]))
            cols = int(g.choice([0, 128, 256]))
            use_ws = bool(g.integers(0, 2))
            A = g.integers(-4, 5, size=(M, K)).astype(np.float32)
            W = g.integers(-2, 3, size=(N, K)).astype(np.float32)
            R = g.integers(-64, 65, size=(M, N)).astype(np.float32)
            ref = round_bf16((A.astype(np.float64) @ W.astype(np.float64).T).astype(np.float32))
            if epi == 1:
                ref = round_bf16(ref + R)
            At = torch.from_numpy(A).cuda().to(torch.bfloat16)
            Wt = torch.from_numpy(W).cuda().to(torch.bfloat16)
            if tiled:
                Wt = tile_major(Wt)
            Rt = torch.from_numpy(R).cuda().to(torch.bfloat16)
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
            ws = torch.empty(lib.llada_gemm_workspace_bytes(M, N, K) + 16, dtype=torch.uint8, device="cuda") if use_ws else None
            lib.llada_gemm_force_tile_rows(rows)
            lib.llada_gemm_force_tile_cols(cols)
            desc = f"it {it}: M={M} N={N} K={K} epi={epi} rows={rows} cols={cols} ws={use_ws} tiled={tiled}"
            _lib.check(lib.llada_gemm_bf16(At.data_ptr(), Wt.data_ptr(), out.data_ptr(), M, N, K, K, N,
                                           epi | (_lib.GEMM_W_TILED if tiled else 0), Rt.data_ptr() if epi == 1 else None,
                                           N if epi == 1 else 0, None, None, 0, _lib.ptr(ws), 0 if ws is None else ws.numel() - 16,
                                           st))
            torch.cuda.synchronize()
            got = out.float().cpu().numpy()
            assert np.array_equal(got, ref), (desc, int((got != ref).sum()))
    finally:
        lib.llada_gemm_force_tile_rows(0)
        lib.llada_gemm_force_tile_cols(0)
    return n_iter


def run_rows(n_iter, seed):
    """Random calls of llada_remask_scores: vocabulary 8 .. 24000 (multiples of 8), all rules, both EOS controls
    with the special ids anywhere in the vocabulary, temperature 0 / 0.7 (injected noise), guidance, active-row
    lists, and every row-kernel form (whole row on chip; streamed pieces: automatic, 1, 3 or 8 per row), against
    the oracle's confidence_step.  x0 must agree wherever it can be used; confidences within one bf16 step."""
    import ctypes as C

    from llada_amd import _lib
    from oracle.bf16 import round_bf16
    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(seed))
    st = torch.cuda.current_stream().cuda_stream
    saved = osamp.EOS_ID, osamp.EOT_ID
    try:
        for it in range(n_iter):
            V = 8 * int(g.integers(1, 3001))
            B, L = int(g.integers(1, 4)), int(g.integers(1, 12))
            rule = RULES[int(g.integers(0, 5))]
            T = float(g.choice([0.0, 0.0, 0.7]))
            flags = int(g.integers(0, 4)) if V >= 16 else 0
            cfg_scale = float(g.choice([0.0, 0.0, 1.5]))
            mode = [None, 0, 1, 3, 8][int(g.integers(0, 5))]       # None: no workspace
            eos, eot = (int(v) for v in g.choice(V, size=2, replace=False)) if V >= 16 else (0, 1)
            osamp.EOS_ID, osamp.EOT_ID = eos, eot
            mask_id = V - 1
            x = g.integers(0, max(1, V - 1), size=(B, L), dtype=np.int64)
            x[g.random((B, L)) < 0.6] = mask_id
            block_end = int(g.integers(1, L + 1))
            scale = float(g.choice([1.0, 3.0, 9.0]))
            cond = round_bf16(g.standard_normal((B, L, V), dtype=np.float32) * np.float32(scale))
            un = round_bf16(g.standard_normal((B, L, V), dtype=np.float32) * np.float32(scale)) if cfg_scale else None
            # plant the seams: repeated maximum, maximum on a special token
            for b in range(B):
                for col in range(L):
                    r = g.random()
                    if r < 0.2:
                        cond[b, col, g.choice(V, size=min(3, V), replace=False)] = np.float32(4 * scale)
                    elif r < 0.4:
                        cond[b, col, eos if r < 0.3 else eot] = np.float32(5 * scale)
            u = g.random((B, L, V)) if T else None
            rs = g.random((B, L)).astype(np.float32)
            logits = osamp.cfg_combine(cond, un, cfg_scale) if cfg_scale else cond.copy()
            ox0, oconf = osamp.confidence_step(x, logits, mask_id, block_end, rule, T, u, rs, bool(flags & 1), bool(flags & 2))
            # active-row list: the masked positions left of block_end plus a few others, or every row
            if g.integers(0, 2):
                live = np.flatnonzero(((x == mask_id) & (np.arange(L)[None, :] < block_end)).reshape(-1))
                extra = g.choice(B * L, size=min(B * L, 2), replace=False)
                rows = np.unique(np.concatenate([live, extra])).astype(np.int32)
            else:
                rows = None
            sel = np.arange(B * L) if rows is None else rows
            ct = torch.from_numpy(cond.reshape(B * L, V)[sel]).cuda().to(torch.bfloat16).contiguous()
            ut = None if un is None else torch.from_numpy(un.reshape(B * L, V)[sel]).cuda().to(torch.bfloat16).contiguous()
            nt = None if u is None else torch.from_numpy(u.reshape(B * L, V)[sel]).cuda().contiguous()
            rt = torch.from_numpy(rs).cuda()
            xt = torch.from_numpy(x).cuda()
            rp = None if rows is None else torch.from_numpy(rows).cuda()
            x0 = torch.full((B, L), -7, dtype=torch.long, device="cuda")
            conf = torch.full((B, L), 123.0, dtype=torch.float32, device="cuda")
            args = _lib.RemaskArgs(ct.data_ptr(), _lib.ptr(ut), cfg_scale, _lib.ptr(rp), ct.shape[0], V, xt.data_ptr(), B, L,
                                   block_end, mask_id, _lib.REMASK[rule], flags, eos, eot, T, _lib.ptr(nt), rt.data_ptr(), 5, 2, 1,
                                   x0.data_ptr(), conf.data_ptr())
            if mode is not None:
                ws = torch.empty(lib.llada_rows_workspace_bytes(ct.shape[0]), dtype=torch.uint8, device="cuda")
                args.workspace, args.workspace_bytes = ws.data_ptr(), ws.numel()
                lib.llada_rows_force_splits(mode)
            desc = (f"it {it}: V={V} B={B} L={L} rule={rule} T={T} flags={flags} cfg={cfg_scale} mode={mode} eos={eos} eot={eot} "
                    f"block_end={block_end} rows={'all' if rows is None else rows.size}")
            _lib.check(lib.llada_remask_scores(C.byref(args), st))
            torch.cuda.synchronize()
            gx0, gconf = x0.cpu().numpy(), conf.cpu().numpy()
            livem = ~np.isneginf(oconf)
            assert np.array_equal(np.isneginf(gconf), np.isneginf(oconf)), desc
            assert np.array_equal(np.isnan(gconf), np.isnan(oconf)), desc
            assert np.array_equal(gx0[livem], ox0[livem]), desc
            assert np.array_equal(gx0[~livem], x[~livem]), desc
            fin = np.isfinite(oconf)
            if rule == "random":
                assert np.array_equal(gconf[fin], oconf[fin]), desc
            else:
                tol = 2.0 ** -8 if rule == "high_confidence" else (4e-7 if "entropy" in rule else 0.0)
                bad = (ulp_distance(gconf[fin], oconf[fin]) > 1) & (np.abs(gconf[fin] - oconf[fin]) > tol)
                assert not bad.any(), (desc, gconf[fin][bad][:4], oconf[fin][bad][:4])
    finally:
        osamp.EOS_ID, osamp.EOT_ID = saved
        lib.llada_rows_force_splits(0)
    return n_iter
