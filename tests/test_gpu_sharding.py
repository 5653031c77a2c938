"""Row (e) on hardware: the batch-shard path of diffroll_amd.distributed driven by the REAL engine.

Only one GPU is leased for the test run, so the N-rank job is covered from both sides:
  * the collective side - a 1-rank RCCL process group (backend 'nccl'): init_process_group with device_id,
    all_gather_into_tensor, all_reduce and barrier execute on the device, through the same sample_sharded /
    gather_rolls code the N-rank job runs;
  * the partition side - G in {2, 3, 8} ranks emulated in turn on the one device through sample_shard()'s own
    slicing (incl. ranks whose shard is empty) and the same pad / un-pad code as the gather: the assembled result
    must equal the unsharded one (Philox is keyed by the global sample index).
`python bench.py --gpus 2` on a 1-GPU box must fail with a device-count message, not a launcher hint.
"""
import os
import subprocess
import sys

import pytest
import torch

from oracle import diffroll_ref as R

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ATOL_SHARD = 1e-5      # other local batch size -> other tile flavour / split-K order: fp32 round-off (observed ~1e-6)


def _model(sampler="cfdg_ddpm_x0", layers=4, steps=12, k=9, C=128):
    from test_gpu_parity import make_model
    hp = dict(R.DEFAULT_HP)
    hp.update(residual_channels=C, residual_layers=layers, kernel_size=k, timesteps=steps)
    p = R.synthetic_params(hp, seed=11)
    return hp, p, make_model(hp, p, sampler=sampler, w=0.5)


@pytest.mark.parametrize("G", [2, 3, 8])
@pytest.mark.parametrize("mode", ["philox", "injected"])
def test_fake_ranks_equal_unsharded(G, mode):
    """B = 5 clips over G ranks: shards of 3+2, 2+2+1 and 1+1+1+1+1+0+0+0 (three empty ranks)."""
    from diffroll_amd.distributed import sample_sharded, sample_sharded_sequential, shard_bounds
    hp, p, m = _model()
    torch.manual_seed(5)
    B, Tn = 5, 40
    wav = 0.1 * torch.randn(B, Tn * 512)
    x = torch.randn(B, 1, Tn, 88)
    noise = torch.randn(hp["timesteps"], B, 1, Tn, 88) if mode == "injected" else None
    whole = sample_sharded(m, x, wav, noise, seed=3)                 # no process group: world = 1
    assert whole.shape == (B, 1, Tn, 88) and bool(torch.isfinite(whole).all())
    parts = sample_sharded_sequential(m, x, wav, noise, seed=3, world_size=G)
    assert parts.shape == whole.shape
    d = float((parts - whole).abs().max())
    assert d <= ATOL_SHARD, d
    sizes = [shard_bounds(B, r, G) for r in range(G)]
    assert sum(hi - lo for lo, hi in sizes) == B and (G != 8 or sum(hi == lo for lo, hi in sizes) == 3)
    if mode == "injected":                                           # and both equal the oracle's chain
        with torch.no_grad():
            ref = R.sample_chain(p, hp, "cfdg_ddpm_x0", x, wav, noise, w=0.5)
        assert float((parts.cpu() - ref).abs().max()) <= 1e-5


def test_fake_ranks_generation_without_waveform():
    from diffroll_amd.distributed import sample_sharded, sample_sharded_sequential
    hp, p, m = _model(sampler="generation_ddpm_x0")
    torch.manual_seed(6)
    x = torch.randn(6, 1, 48, 88)
    whole = sample_sharded(m, x, None, seed=1)
    parts = sample_sharded_sequential(m, x, None, seed=1, world_size=4)
    assert float((parts - whole).abs().max()) <= ATOL_SHARD


_RANK_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch
from diffroll_amd import launch
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist = launch.init_process_group(dev, force_single=True)         # 1-rank RCCL group
info = launch.dist_info(dist)
assert info["ranks_seen"] == 1 and info["backend"] == "nccl", info
from test_gpu_sharding import _model
from diffroll_amd.distributed import sample_sharded, gather_rolls, world
assert world() == (0, 1)
hp, p, m = _model()
torch.manual_seed(5)
B, Tn = 3, 40
wav = 0.1 * torch.randn(B, Tn * 512)
x = torch.randn(B, 1, Tn, 88)
a = sample_sharded(m, x, wav, seed=3)                            # through all_gather_into_tensor (1 rank)
roll, _ = m.sample(x, wav, seed=3)
assert torch.equal(a, roll), float((a - roll).abs().max())
g = gather_rolls(roll)
assert torch.equal(g, roll)
t = torch.ones(4, device=dev)
dist.all_reduce(t)
dist.barrier()
torch.cuda.synchronize()
assert float(t.sum()) == 4.0
dist.destroy_process_group()
print("RANK_OK " + json.dumps(info))
"""


def test_one_rank_rccl_group_runs_the_collective_path():
    """init_process_group('nccl', device_id=...) + all_gather_into_tensor + all_reduce + barrier on the device, in a
    child process (a process group is process-global state)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", _RANK_SCRIPT.format(root=ROOT)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_one_rank_under_torch_distributed_run():
    """The driver's own launch line with one local rank: python -m torch.distributed.run --nproc-per-node 1
    bench.py --gpus 1 (config 1: the 50-step single clip, seconds) - the JSON line reports the RCCL group."""
    import json
    from diffroll_amd.launch import free_port
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--config", "1",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-split", "--no-roofline", "--no-cold-start"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["dist"]["ranks_seen"] == 1 and j["dist"]["backend"] == "nccl", j["dist"]
    assert j["value"] > 0 and j["config"]["baseline_config"] == 1
    # what the engine launched is part of the line (dr_launch_state): a single clip runs one launch per phase, nobody yielded
    assert j["fused_yields"] == 0 and j["fused_fallbacks"] == 0 and j["per_rank_launch_mode"] == ["per_phase"] and j["launch_mode"] == "per_phase"


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0
    msg = r.stdout + r.stderr
    assert f"only {n} HIP device" in msg and "torch.distributed.run" not in msg.split("only")[0][-200:], msg[-1000:]


def test_native_comm_one_rank_gather_through_the_cabi():
    """dr_comm_unique_id -> dr_comm_create (ncclCommInitRank, 1 rank) -> dr_gather (ncclAllGather) on the device, no
    torch.distributed anywhere: the gathered tensor equals the shard; the library reports its RCCL version."""
    from diffroll_amd.distributed import NativeComm, gather_rolls
    dev = torch.device("cuda", 0)
    comm = NativeComm(dev, rank=0, world_size=1)
    assert comm.rccl_version() > 20000
    x = torch.randn(5, 1, 37, 88, device=dev)
    y = gather_rolls(x, comm=comm)
    torch.cuda.synchronize()
    assert y.shape == x.shape and torch.equal(y, x)
    z = comm.all_gather(x[:, 0].contiguous())
    assert torch.equal(z, x[:, 0])
    comm.close()


_TWO_RANK_SCRIPT = r"""
import os, sys, json
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch
import torch.distributed as dist
rank, world = int(sys.argv[1]), int(sys.argv[2])
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = sys.argv[3]
torch.cuda.set_device(0)                     # both ranks on the one leased GPU
dist.init_process_group("gloo", rank=rank, world_size=world)
from test_gpu_sharding import _model
from diffroll_amd.distributed import sample_sharded, world as wfn
assert wfn() == (rank, world)
hp, p, m = _model(layers=3, steps=8, C=512)       # full width, 14 evaluations per rank: the fused residual-stack kernel is the one that runs
torch.manual_seed(5)
B, Tn = 14, 125
wav = 0.1 * torch.randn(B, Tn * 512)
x = torch.randn(B, 1, Tn, 88)
out = []
for rep in range(3):
    full = sample_sharded(m, x, wav, seed=3 + rep)
    out.append(full.cpu())
flag, _ = m.engine.stack_status()
dist.barrier()
if rank == 0:
    torch.save(out, sys.argv[4])
print("RANK_DONE", rank, flag, m.engine.stack_launches, m.engine.fallbacks)
dist.destroy_process_group()
"""


def test_two_processes_share_the_gpu_and_gather(tmp_path):
    """A real 2-process job on the ONE leased GPU (gloo rendezvous, host-side gather): each rank runs its 7-clip shard
    through its own engine - the processes find each other in the driver's process list and stop fusing (no
    time-out, round 5) - and every rank returns the full
    batch, equal to the unsharded result of a single process.  (The RCCL flavour of the same job needs two GPUs.)"""
    from diffroll_amd.distributed import sample_sharded
    from diffroll_amd.launch import free_port
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    port = str(free_port())
    res = str(tmp_path / "full.pt")
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_RANK_SCRIPT.format(root=ROOT), str(r), "2", port, res], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0 and "RANK_DONE" in so, (so[-1500:], se[-3000:])
        done = [ln for ln in so.splitlines() if ln.startswith("RANK_DONE")][-1].split()
        assert done[2] == "0" and done[4] == "0", so    # no barrier time-out in either process, pending or healed: a process
        #                                                 that sees the other one computing (csrc/tenants.h) yields to per-phase
        #                                                 launches instead of running into the ~1 s spin bound
    got = torch.load(res)
    hp, p, m = _model(layers=3, steps=8, C=512)
    torch.manual_seed(5)
    B, Tn = 14, 125
    wav = 0.1 * torch.randn(B, Tn * 512)
    x = torch.randn(B, 1, Tn, 88)
    for rep in range(3):
        whole = sample_sharded(m, x, wav, seed=3 + rep).cpu()
        d = float((got[rep] - whole).abs().max())
        assert d <= ATOL_SHARD, (rep, d)


def _clean_env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "DR_BENCH_SHARE_GPU", "DR_SELF_SPAWNED"):
        env.pop(k, None)
    return env


def test_two_rank_bench_path_executes_on_one_gpu():
    """`bench.py --gpus 2 --share-gpu`: the N > 1 benchmark path - launch.spawn_ranks ->
    torch.distributed.run -> init_process_group -> barrier -> timed loop -> all_reduce(MAX) -> per-rank all_gather ->
    gather_rolls -> the JSON line - with both ranks on the one leased device over gloo.  No hardware claim follows from
    it; it guarantees that the real 8-GPU run does not die in plumbing."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--config", "1",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-split", "--no-roofline", "--no-cold-start"],
                       env=_clean_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dist"]["ranks_seen"] == 2 and j["dist"]["backend"] == "gloo" and j["dist"]["share_gpu"] is True
    assert len(j["per_rank_ms_per_step"]["all"]) == 2 and min(j["per_rank_ms_per_step"]["all"]) > 0
    assert j["scaling"] == "weak" and j["steps"] == 2 and j["config"]["parallelism"] == "batch-shard x2"
    # value = BOTH ranks' frames over the max-over-ranks time
    assert abs(j["value"] - 2 * 1 * 125 * 1e3 / j["ms_per_step"]) <= 0.01 * j["value"]
    assert j["gather_us"] > 0
    assert j["per_rank_launch_mode"] == ["per_phase", "per_phase"]      # one entry per rank (--share-gpu asks for per-phase launches)


def _needs_fused_default(fn):
    """(a forced-mode run of the suite with DR_TEST_TUNE="fused_stack=0" has nothing to yield: per-phase launches are the default there)"""
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        from tools import tuning_env
        if tuning_env.forced("fused_stack", 1) == 0:
            pytest.skip("fused launches are switched off for this run (DR_TEST_TUNE)")
        return fn(*a, **k)
    return wrapper


def _fake_kfd_tree(root, busy):
    """A copy of this box's KFD topology with a made-up process list: every GPU has two queue holders, one of them computing
    (busy) or none.  Returns False where the real tree is not readable (no sysfs in the container)."""
    real = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(real):
        return False
    gids = []
    for nd in os.listdir(real):
        try:                                             # (inside a 1-GPU lease only the leased GPU's node is readable)
            gid = open(f"{real}/{nd}/gpu_id").read().strip()
            props = open(f"{real}/{nd}/properties").read()
        except OSError:
            continue
        d = os.path.join(root, "topology", "nodes", nd)
        os.makedirs(d)
        open(os.path.join(d, "gpu_id"), "w").write(gid + "\n")
        open(os.path.join(d, "properties"), "w").write(props)
        if gid != "0":
            gids.append(gid)
    for pid, occ in ((4001, 0), (4002, 133 if busy else 0)):
        for n, gid in enumerate(gids):
            qd = os.path.join(root, "proc", str(pid), "queues", str(n))
            os.makedirs(qd)
            open(os.path.join(qd, "gpuid"), "w").write(gid + "\n")
            sd = os.path.join(root, "proc", str(pid), f"stats_{gid}")
            os.makedirs(sd)
            open(os.path.join(sd, "cu_occupancy"), "w").write(f"{occ}\n")
    return bool(gids)


@_needs_fused_default
def test_engine_yields_to_a_busy_co_tenant_and_comes_back(tmp_path):
    """VERDICT r5 item 1 / ADVICE r5: a yield is VISIBLE (dr_launch_state: yields, mode, fused_enabled) and RECOVERABLE.
    The engine is shown a KFD process list in which another process computes on its GPU (dr_debug_kfd_root): it yields at
    creation and samples with one launch per phase; shown a list in which the co-holder idles, two looks in front of later
    chains switch the fused launches back on.  Same rolls all along."""
    import time
    from diffroll_amd import _cabi
    busy, idle = str(tmp_path / "busy"), str(tmp_path / "idle")
    if not (_fake_kfd_tree(busy, True) and _fake_kfd_tree(idle, False)):
        pytest.skip("no readable /sys/class/kfd/kfd/topology in this container")
    lib = _cabi.load_library()
    import gc
    gc.collect()                                          # engines of earlier tests
    lib.dr_debug_kfd_root(busy.encode())
    try:
        hp, p, m = _model(layers=3, steps=8, C=512)
        torch.manual_seed(11)
        B, Tn = 16, 125                                   # 32 evaluations x 8 M tiles: the fused kernels' own geometry
        wav = 0.1 * torch.randn(B, Tn * 512)
        x = torch.randn(B, 1, Tn, 88)
        eng = m.engine
        st = eng.launch_state()
        # (the only engine of the process on the device decides at creation; with engines of earlier tests still alive the
        # look at creation may not wait for THEIR work and stays undecided - the look in front of the first chain decides)
        assert st["yields"] in (0, 1) and st["fused_enabled"] == 1 - st["yields"] and eng.yields == st["yields"], st
        a = m.sample(x, wav, seed=2)[0]
        st = eng.launch_state()
        assert st["mode"] == "per_phase" and st["yields"] == 1 and st["fallbacks"] == 0 and st["rearms"] == 0, st
        time.sleep(0.3)
        m.sample(x, wav, seed=2)                          # a look that still finds the tenant computing: nothing changes
        assert eng.launch_state()["fused_enabled"] == 0
        lib.dr_debug_kfd_root(idle.encode())              # the tenant has stopped computing (it still holds its queue)
        for _ in range(2):
            time.sleep(0.3)
            m.sample(x, wav, seed=2)
        st = eng.launch_state()
        assert st["rearms"] == 1 and st["fused_enabled"] != 0, st
        b = m.sample(x, wav, seed=2)[0]
        st = eng.launch_state()
        assert st["mode"] == "fused_stack+tail" and st["yields"] == 1 and st["fallbacks"] == 0, st
        assert float((a - b).abs().max()) <= ATOL_SHARD
    finally:
        lib.dr_debug_kfd_root(None)


@_needs_fused_default
def test_bench_refuses_to_print_a_line_when_the_engine_yielded(tmp_path):
    """... and a yield is FATAL for a measurement: bench.py, shown the same busy co-tenant (DR_BENCH_FAKE_KFD), exits
    non-zero without a JSON line instead of reporting per-phase launches as the engine's throughput - the one way the
    first 8-GPU scaling run could have come out wrong without anything failing."""
    busy = str(tmp_path / "busy")
    if not _fake_kfd_tree(busy, True):
        pytest.skip("no readable /sys/class/kfd/kfd/topology in this container")
    env = _clean_env()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
           "--no-split", "--no-roofline", "--no-cold-start"]
    r = subprocess.run(cmd, env=dict(env, DR_BENCH_FAKE_KFD=busy), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode != 0 and '"metric"' not in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
    assert "yield" in r.stderr and "no benchmark line" in r.stderr, r.stderr[-2000:]
    # the same command on the real process list: a line, with the record of what was launched
    import json
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1])
    assert j["fused_yields"] == 0 and j["fused_fallbacks"] == 0 and j["launch_mode"] == "fused_stack+tail", j
    assert j["attempts"] == 1 and j["discarded_attempts"] == [] and j["yields_since_creation"] == [0], j


@_needs_fused_default
def test_bench_repeats_a_discarded_attempt_once_the_co_tenant_is_gone(tmp_path):
    """A timed region that started on a yielded engine is discarded by all ranks and repeated: once the made-up co-tenant
    idles (DR_BENCH_FAKE_KFD_THEN), the repeat's warm-up takes the two clean looks that switch the fused launches back on,
    and the line that is printed says so: attempts 2, one discarded attempt, no yield inside the accepted timed region."""
    import json
    busy, idle = str(tmp_path / "busy"), str(tmp_path / "idle")
    if not (_fake_kfd_tree(busy, True) and _fake_kfd_tree(idle, False)):
        pytest.skip("no readable /sys/class/kfd/kfd/topology in this container")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "3", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
           "--no-split", "--no-roofline", "--no-cold-start"]
    r = subprocess.run(cmd, env=dict(_clean_env(), DR_BENCH_FAKE_KFD=busy, DR_BENCH_FAKE_KFD_THEN=idle), capture_output=True,
                       text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    assert "attempt 1 of 3 discarded" in r.stderr, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1])
    assert j["attempts"] == 2 and len(j["discarded_attempts"]) == 1 and j["discarded_attempts"][0]["per_rank_launch_mode"] == ["per_phase"], j
    assert j["fused_yields"] == 0 and j["fused_fallbacks"] == 0 and j["launch_mode"] == "fused_stack+tail", j
    assert j["yields_since_creation"] == [1], j


def test_scale_table_emits_a_scale_record_for_one_and_two_ranks(tmp_path):
    """tools/scale_table.py --gpus 1,2 --share-gpu --scale-json: the SCALE-shaped record the 8-GPU day will produce,
    from a 1-rank run and a 2-rank run that really had two ranks (ranks_seen checked per cell; exit code 0)."""
    import json
    out = str(tmp_path / "scale.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scale_table.py"), "--gpus", "1,2", "--configs", "1",
                        "--steps", "2", "--warmup", "1", "--share-gpu", "--scale-json", out],
                       env=_clean_env(), capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    rec = json.load(open(out))
    assert rec["ok"] is True and rec["problems"] == []
    rows = rec["configs"]["1"]
    assert [row["n_gpus"] for row in rows] == [1, 2] and [row["ranks_seen"] for row in rows] == [1, 2]
    assert rows[0]["share_gpu"] is False and rows[1]["share_gpu"] is True and len(rows[1]["per_rank_ms_all"]) == 2
    assert rows[1]["efficiency_vs_n1"] is not None and rows[1]["value"] > 0


def test_bench_exits_non_zero_when_the_group_is_smaller_than_asked():
    """One rank started by hand with WORLD_SIZE=1 but --gpus 2: refused, no JSON line."""
    env = _clean_env()
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
    from diffroll_amd.launch import free_port
    env["MASTER_PORT"] = str(free_port())
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--config", "1"],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and '"metric"' not in r.stdout


def test_external_launcher_line_with_two_ranks_on_one_gpu():
    """The driver's own launch line - python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ... - with
    --share-gpu appended (both ranks on device 0): the ranks JOIN the launcher's group instead of spawning their own."""
    import json
    from diffroll_amd.launch import free_port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--config", "1",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-split", "--no-roofline", "--no-cold-start"]
    r = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dist"]["ranks_seen"] == 2 and j["dist"]["launcher"] == "torch.distributed.run"
    assert len(j["per_rank_ms_per_step"]["all"]) == 2
