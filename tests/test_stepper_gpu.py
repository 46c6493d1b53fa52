"""llada_amd.stepper.GenerationStepper (the loop body of generate.py:80-134 offered piecewise, as the interactive
visualiser drives it: viz/generation_engine.py:160-341): driven with the reference's schedule it reproduces
generate() token for token; committing from cached scores, caller-forced tokens and rewinding behave as the
visualiser's engine defines them; its scores are the oracle's."""
import numpy as np
import pytest

from oracle import sampler as osamp
from tests import fuzz_util

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

MASK_ID = fuzz_util.MASK_ID


@pytest.fixture(scope="module")
def built():
    return fuzz_util.build_model()


def schedule(n, steps):
    base, rem = divmod(n, steps)
    return [base + (1 if i < rem else 0) for i in range(steps)]


@pytest.mark.parametrize("temperature,remasking,cfg", [(0.0, "low_confidence", 0.0), (0.7, "low_confidence", 0.0),
                                                       (0.0, "random", 0.0), (0.7, "high_entropy", 1.5)])
def test_stepper_with_the_reference_schedule_is_generate(built, temperature, remasking, cfg):
    from llada_amd import generate
    from llada_amd.stepper import GenerationStepper
    _, _, model = built
    g = np.random.Generator(np.random.PCG64(5))
    B, P, G, bl, steps = 3, 11, 32, 16, 12                      # 2 blocks x 6 steps: 3,3,3,3,2,2 tokens per step
    prompt = torch.from_numpy(g.integers(0, MASK_ID, size=(B, P), dtype=np.int64))
    am = torch.ones(B, P, dtype=torch.long) if cfg > 0 else None
    want = generate(model, prompt, am, steps=steps, gen_length=G, block_length=bl, temperature=temperature, cfg_scale=cfg,
                    remasking=remasking, mask_id=MASK_ID, seed=21)
    st = GenerationStepper(model, prompt, am, gen_length=G, block_length=bl, temperature=temperature, cfg_scale=cfg,
                           remasking=remasking, mask_id=MASK_ID, seed=21)
    for blk in range(G // bl):
        assert st.block == blk and not st.done
        for k in schedule(bl, steps // (G // bl)):
            st.step(k)
    assert st.done and torch.equal(st.x, want)
    with pytest.raises(StopIteration):
        st.forward()


def test_select_from_cached_scores_manual_tokens_and_rewind(built):
    from llada_amd.stepper import GenerationStepper
    ocfg, W, model = built
    g = np.random.Generator(np.random.PCG64(6))
    P, G, bl = 9, 16, 8
    prompt = g.integers(0, MASK_ID, size=(1, P), dtype=np.int64)
    st = GenerationStepper(model, torch.from_numpy(prompt), gen_length=G, block_length=bl, mask_id=MASK_ID)
    with pytest.raises(ValueError):
        st.select(1)
    x0, conf = st.forward()
    x0, conf = x0.cpu().numpy().copy(), conf.cpu().numpy().copy()
    # the scores are the oracle's epilogue on the HIP model's own logits
    x_in = st.x.cpu().numpy().copy()
    lg = model(st.x).logits.float().cpu().numpy()
    ox0, oconf = osamp.confidence_step(x_in, lg, MASK_ID, P + bl, "low_confidence")
    live = ~np.isneginf(oconf)
    assert np.array_equal(live, ~np.isneginf(conf)) and np.array_equal(x0[live], ox0[live])
    assert np.abs(conf[live] - oconf[live]).max() <= 2.0 ** -7
    # two selections of one token on the same scores == the two most confident candidates
    st.select(1)
    st.select(1)
    want = x_in.copy()
    osamp.topk_transfer(want, x0, conf, np.array([2]))
    assert np.array_equal(st.x.cpu().numpy(), want)
    # a forced token: the caller's choice wins, and it is no candidate any more
    free = int(np.flatnonzero(st.x.cpu().numpy()[0] == MASK_ID)[0])
    st.select(0, manual_selections={free: 7})
    assert int(st.x[0, free]) == 7 and (st.x == MASK_ID).sum() == G - 3
    # asking for more than the block has left commits what is left of the block only
    st.select(100)
    assert (st.x[0, P:P + bl] != MASK_ID).all() and (st.x[0, P + bl:] == MASK_ID).all() and st.block == 1
    # rewind: back to the state after the first selection, scores must be recomputed
    states = len(st.history)
    assert states == 5
    back = st.rewind(1)
    assert (back == MASK_ID).sum() == G - 1 and len(st.history) == 2 and st.block == 0
    with pytest.raises(ValueError):
        st.select(1)
    assert st.rewind(17) is None
    st.step(2)
    assert (st.x == MASK_ID).sum() == G - 3


def test_rewind_restores_the_random_stream_and_forced_ids_are_checked(built):
    """Sampled runs (temperature 0.7, library Philox noise keyed by the forward counter): stepping on after a
    rewind repeats the original continuation exactly; a forced token outside the vocabulary raises like a prompt
    id would; ``max_history`` bounds the kept states."""
    from llada_amd.stepper import GenerationStepper
    _, _, model = built
    g = np.random.Generator(np.random.PCG64(8))
    P, G, bl = 7, 16, 8
    prompt = torch.from_numpy(g.integers(0, MASK_ID, size=(2, P), dtype=np.int64))
    st = GenerationStepper(model, prompt, gen_length=G, block_length=bl, temperature=0.7, mask_id=MASK_ID, seed=13)
    states = [st.x.clone()]
    for _ in range(6):
        states.append(st.step(2).clone())
    assert st.step_index == 6
    st.rewind(2)
    assert st.step_index == 2 and torch.equal(st.x, states[2]) and sorted(st.history) == [0, 1, 2]
    for n in range(3, 7):
        assert torch.equal(st.step(2), states[n]), n              # same noise as the first time round
    with pytest.raises(IndexError):
        st.select(0, manual_selections={(0, P): model.cfg.vocab_size})
    with pytest.raises(IndexError):
        st.step(1, manual_selections={(5, P): 3})                 # row outside the batch
    capped = GenerationStepper(model, prompt, gen_length=G, block_length=bl, mask_id=MASK_ID, max_history=2)
    for _ in range(5):
        capped.step(1)
    assert sorted(capped.history) == [0, 4, 5]
    assert capped.rewind(2) is None and capped.rewind(4) is not None and capped.step_index == 4
