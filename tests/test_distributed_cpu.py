"""The N>1 path on CPU: world_size 2, gloo backend.  The shard -> per-rank chain -> gather logic of
diffroll_amd.distributed is exercised with a stand-in model whose ``sample`` is a deterministic
per-sample function (the real engine needs a GPU); the result must equal the single-process one and
must not depend on the world size."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


class FakeEngine:
    device = torch.device("cpu")


class FakeModel:
    """sample() mimics the contract of ClassifierFreeDiffRoll.sample: per-sample independent, noise
    either injected or derived from (seed, global sample index)."""
    engine = FakeEngine()

    def sample(self, x_T, waveform=None, noise=None, seed=0, first_sample=0, use_graph=True):
        B = x_T.shape[0]
        out = x_T.clone() * 0.5
        if waveform is not None:
            out = out + waveform.mean(dim=1).view(B, 1, 1, 1)
        for b in range(B):
            if noise is not None:
                out[b] += noise[:, b].sum(0)
            else:
                g = torch.Generator().manual_seed(seed * 1000003 + first_sample + b)
                out[b] += torch.randn(out[b].shape, generator=g)
        return out, None


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, use_noise, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from diffroll_amd.distributed import gather_rolls, sample_sharded, world as wfn
    assert wfn() == (rank, world)
    torch.manual_seed(0)
    x = torch.randn(B, 1, 6, 88)
    wav = torch.randn(B, 64)
    noise = torch.randn(4, B, 1, 6, 88) if use_noise else None
    full = sample_sharded(FakeModel(), x, wav, noise, seed=5)
    # equal-size gather
    g = gather_rolls(torch.full((2, 1, 3, 88), float(rank)))
    assert g.shape == (2 * world, 1, 3, 88) and all(float(g[2 * r].mean()) == r for r in range(world))
    if rank == 0:
        ret.put(full)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B,use_noise", [(6, True), (7, False), (1, False)])
def test_sharded_sampling_equals_single_process(B, use_noise):
    from diffroll_amd.distributed import sample_sharded
    torch.manual_seed(0)
    x = torch.randn(B, 1, 6, 88)
    wav = torch.randn(B, 64)
    noise = torch.randn(4, B, 1, 6, 88) if use_noise else None
    single = sample_sharded(FakeModel(), x, wav, noise, seed=5)      # no process group: world = 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, use_noise, q)) for r in range(2)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert torch.equal(full, single)
