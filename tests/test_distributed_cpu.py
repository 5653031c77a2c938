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

    def output_frames(self, T, waveform_samples):
        return T

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


@pytest.mark.parametrize("G", [2, 3, 8])
def test_sequential_rank_emulation_equals_single_process(G):
    """sample_sharded_sequential (what the GPU tests use to cover G ranks on one device) walks the same shard /
    pad / un-pad code as the collective path: with the stand-in model it must reproduce the unsharded result
    exactly, empty shards included (B = 5 over 8 ranks)."""
    from diffroll_amd.distributed import pad_shard, sample_sharded, sample_sharded_sequential, shard_bounds, unpad_gathered
    torch.manual_seed(1)
    B = 5
    x = torch.randn(B, 1, 6, 88)
    wav = torch.randn(B, 64)
    noise = torch.randn(4, B, 1, 6, 88)
    for nz in (noise, None):
        single = sample_sharded(FakeModel(), x, wav, nz, seed=9)
        assert torch.equal(sample_sharded_sequential(FakeModel(), x, wav, nz, seed=9, world_size=G), single)
    # the pad / un-pad pair is the identity on any partition
    parts = [x[slice(*shard_bounds(B, r, G))] for r in range(G)]
    assert torch.equal(unpad_gathered(torch.cat([pad_shard(p, B, G) for p in parts], 0), B, G), x)


def test_bench_gpus_flag_spawns_or_refuses_with_a_device_count_message():
    """`python bench.py --gpus N` starts its own ranks; with fewer than N GPUs visible (here: none) it must say so -
    a device-count message, not a hint to wrap the command in a launcher."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: covered by tests/test_gpu_sharding.py")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300, cwd=root)
    assert r.returncode != 0
    assert "2 GPUs requested but only 0 HIP device(s) visible" in r.stdout + r.stderr


def test_launch_helpers():
    from diffroll_amd import launch
    env_backup = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    try:
        assert not launch.under_launcher() and launch.rank_env() == (0, 1, 0)
        os.environ.update(RANK="3", WORLD_SIZE="8", LOCAL_RANK="3")
        assert launch.under_launcher() and launch.rank_env() == (3, 8, 3)
    finally:
        for k, v in env_backup.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    assert launch.dist_info(None)["ranks_seen"] == 1
    p = launch.free_port()
    assert 1024 < p < 65536


def test_scale_record_flags_a_cell_that_ran_with_fewer_ranks_than_asked(monkeypatch):
    """tools/scale_table.py: the SCALE-shaped record marks share-gpu cells, computes efficiency against N = 1, and is NOT ok
    (exit code 1) when a cell's process group had fewer ranks than the N it was asked for."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("scale_table", os.path.join(root, "tools", "scale_table.py"))
    st = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(st)

    def cell(n, value, seen, share=False):
        return {"value": value, "unit": "frames/s", "n_gpus": n, "ms_per_step": 1.0, "scaling": "weak", "_wall_s": 1.0,
                "per_rank_ms_per_step": {"min": 1.0, "max": 1.1, "all": [1.0] * seen}, "gather_us": 3.0,
                "dist": {"ranks_seen": seen, "backend": "nccl", "rccl_version": "2.26.6", "launcher": "x", "share_gpu": share},
                "config": {"workload": "w"}}
    table = {"config2/gpus1": cell(1, 100.0, 1), "config2/gpus2": cell(2, 190.0, 2, share=True),
             "config2/gpus4": {"error": "4 GPUs requested but only 1 HIP device(s) visible"}}
    rec = st.scale_record(table, [1, 2, 4], [2])
    rows = rec["configs"]["2"]
    assert rec["ok"] and rows[1]["efficiency_vs_n1"] == pytest.approx(0.95) and rows[1]["share_gpu"] is True
    assert rows[2]["skipped"] and "only 1 HIP device" in rows[2]["reason"]
    table["config2/gpus2"] = cell(2, 190.0, 1)            # asked for 2, the group had 1
    rec = st.scale_record(table, [1, 2, 4], [2])
    assert not rec["ok"] and "asked for 2 ranks" in rec["problems"][0]
    # ranks that did not all launch the same kernels (one of them yielded to per-phase launches: dr_launch_state) are no
    # scaling point either - bench.py refuses to print such a line, and a record that carries one anyway is flagged
    good = dict(cell(2, 190.0, 2), per_rank_launch_mode=["fused_stack+tail"] * 2, launch_mode="fused_stack+tail", fused_yields=0, fused_fallbacks=0)
    table["config2/gpus2"] = good
    rec = st.scale_record(table, [1, 2, 4], [2])
    assert rec["ok"] and rec["configs"]["2"][1]["per_rank_launch_mode"] == ["fused_stack+tail"] * 2
    table["config2/gpus2"] = dict(good, per_rank_launch_mode=["fused_stack+tail", "per_phase"], launch_mode="mixed", fused_yields=1)
    rec = st.scale_record(table, [1, 2, 4], [2])
    assert not rec["ok"] and "did not all run the same kernels" in rec["problems"][0]


def test_share_gpu_mode_switches_the_group_to_gloo_and_host_tensors(monkeypatch):
    from diffroll_amd import launch
    monkeypatch.delenv("DR_BENCH_SHARE_GPU", raising=False)
    assert not launch.share_gpu()
    monkeypatch.setenv("DR_BENCH_SHARE_GPU", "1")
    assert launch.share_gpu()
    assert launch.collective_device(None, torch.device("cuda", 0)) == torch.device("cpu")

    class _Gloo:
        @staticmethod
        def get_backend():
            return "gloo"

    class _Nccl(_Gloo):
        @staticmethod
        def get_backend():
            return "nccl"
    assert launch.collective_device(_Gloo, torch.device("cuda", 0)) == torch.device("cpu")
    assert launch.collective_device(_Nccl, torch.device("cuda", 0)) == torch.device("cuda", 0)


def test_bench_attempt_verdict_is_collective_and_spares_deliberate_per_phase_runs():
    """bench.py's discard rule, as pure logic: a timed region counts only if NO rank healed a time-out or
    yielded inside it and every rank that was meant to fuse did so at both ends; an engine whose fused launches were switched
    off on purpose (A/B) is not degraded; --share-gpu runs are exempt."""
    import bench
    code = {"per_phase": 1, "fused_stack": 2, "fused_stack+tail": 3, "none": 0}
    clean = dict(fused_enabled=1, fallbacks=0, yields=0, mode="fused_stack+tail")
    good = bench.rank_state(clean, clean, True, code)
    assert good == [0, 0, 3, 1, 0] and bench.degraded_ranks([good] * 8, False) == []
    # a yield inside the timed region on rank 5 of 8: everybody discards
    y = bench.rank_state(clean, dict(clean, yields=1, fused_enabled=0, mode="per_phase"), True, code)
    assert y[:4] == [0, 1, 1, 0]
    bad = bench.degraded_ranks([good] * 5 + [y] + [good] * 2, False)
    assert [r for r, _ in bad] == [5]
    # yielded at creation, never came back: nothing moves inside the timed region, still no measurement
    stuck = dict(fused_enabled=0, fallbacks=0, yields=1, mode="per_phase")
    assert bench.degraded_ranks([bench.rank_state(stuck, stuck, True, code)], False) != []
    # yielded in the warm-up and re-armed before the timed region: fine (yields since creation stay visible)
    back = dict(fused_enabled=1, fallbacks=0, yields=1, mode="fused_stack+tail")
    w = bench.rank_state(back, back, True, code)
    assert bench.degraded_ranks([w], False) == [] and w[4] == 1
    # a healed time-out inside the region
    t = bench.rank_state(clean, dict(clean, fallbacks=1, fused_enabled=0, mode="per_phase"), True, code)
    assert bench.degraded_ranks([good, t], False)[0][0] == 1
    # fused launches switched off on purpose / --share-gpu
    off = dict(fused_enabled=0, fallbacks=0, yields=0, mode="per_phase")
    assert bench.degraded_ranks([bench.rank_state(off, off, False, code)], False) == []
    assert bench.degraded_ranks([y, t], True) == []
