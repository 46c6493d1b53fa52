"""First contact with RCCL on the GPU box: a one-rank "nccl" process group (torch.distributed's nccl backend is
RCCL on ROCm) created inside this process, and the two data-parallel wrappers run through its real collectives on
device tensors -- generate_data_parallel's all-gather of the int64 outputs and get_log_likelihood's all-reduce of
the float64 loss sums -- against the same calls without a group.  One rank exchanges nothing, but the library is
loaded, a communicator is built on the device, and the collectives are enqueued on device buffers exactly as the
8-rank run enqueues them (llada_amd/parallel.py).  The multi-rank partition logic is covered by the world-size-2
gloo test on the CPU and by the rank-by-rank tests of tests/test_configs_gpu.py.

The group lives in the test process (no child is started: a process that has initialised the GPU must not
exec another program on this pool).  Collected late on purpose: a communicator problem must not hide the
other tests."""
import socket

import numpy as np
import pytest

from tests import fuzz_util

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

MASK_ID = fuzz_util.MASK_ID


@pytest.fixture(scope="module")
def rccl_group():
    import torch.distributed as dist
    if not dist.is_nccl_available():
        pytest.fail("torch.distributed has no nccl (RCCL) backend")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist.group.WORLD
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_wrappers_through_a_one_rank_rccl_group(rccl_group):
    import torch.distributed as dist

    from llada_amd import generate, get_log_likelihood
    from llada_amd.parallel import generate_data_parallel
    assert dist.get_backend(rccl_group) == "nccl" and dist.get_world_size(rccl_group) == 1
    # the collectives themselves, on device tensors
    t = torch.arange(6, dtype=torch.float64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=rccl_group)
    got = [torch.empty(3, 5, dtype=torch.long, device="cuda")]
    src = torch.arange(15, dtype=torch.long, device="cuda").reshape(3, 5)
    dist.all_gather(got, src, group=rccl_group)
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(6, dtype=torch.float64)) and torch.equal(got[0], src)

    _, _, model = fuzz_util.build_model()
    g = np.random.Generator(np.random.PCG64(12))
    prompt = torch.from_numpy(g.integers(0, MASK_ID, size=(3, 9), dtype=np.int64))
    kw = dict(steps=8, gen_length=16, block_length=8, temperature=0.7, remasking="low_confidence", mask_id=MASK_ID, seed=4)
    plain = generate(model, prompt, None, **kw)
    through = generate_data_parallel(generate, model, prompt, None, process_group=rccl_group, **kw)
    assert through.is_cuda and torch.equal(through, plain)
    ans = torch.from_numpy(g.integers(0, MASK_ID, size=7, dtype=np.int64))
    a = get_log_likelihood(model, prompt[0], ans, mc_num=8, batch_size=4, mask_id=MASK_ID, seed=2)
    b = get_log_likelihood(model, prompt[0], ans, mc_num=8, batch_size=4, mask_id=MASK_ID, seed=2, rank=0, world_size=1,
                           process_group=rccl_group)
    assert a == b and np.isfinite(a) and a < 0
