"""Child of tests/test_gpu_r3.py::test_timeout_recovery_on_the_hook_build: runs under DR_LIB=<the "hook" variant>, the
only build of the library that knows the option "stack_fault_test" (-DDR_FAULT_HOOK: the persistent kernels' group barriers
can be told to wait for one arrival too many, so that the first wait of a fused launch runs into its spin bound).  Not
collected by a plain `pytest tests` (the file name does not match): the production library has no fault injection.

  * a fused-kernel barrier time-out can never hand out a wrong roll: the consume points (sample / predict_step / the
    samplers / forward) verify and re-run on the per-phase kernels (include/diffroll_amd.h: dr_finish, dr_sample_checked);
  * while a time-out is pending every computing / consuming entry point refuses;
  * dr_gather reports an invalid shard on EVERY rank: the status word travels with the rolls.
"""
import time

import pytest
import torch

from oracle import diffroll_ref as R
from test_gpu_parity import ATOL_STEP, make_model, maxdiff

pytestmark = pytest.mark.gpu


def test_this_is_the_hook_build():
    import os
    from diffroll_amd import _cabi
    assert "hook" in os.path.basename(_cabi.LIB_PATH), "run through tests/test_gpu_r3.py (DR_LIB = the hook variant)"


def _timeout_fixture():
    hp = dict(R.DEFAULT_HP)
    hp.update(residual_channels=128, residual_layers=2, kernel_size=3, timesteps=4)
    p = R.synthetic_params(hp, seed=1)
    m = make_model(hp, p, sampler="generation_ddpm_x0", w=0.0)
    torch.manual_seed(0)
    x = torch.randn(8, 1, 64, 88)
    noise = torch.randn(4, 8, 1, 64, 88)
    with torch.no_grad():
        ref = R.sample_chain(p, hp, "generation_ddpm_x0", x, None, noise)
    return m, x, noise, ref


def test_sample_with_a_timed_out_fused_launch_returns_the_right_roll():
    """stack_fault_test = 1 makes the first group barrier of every fused launch run into its spin bound.  ONE call of
    m.sample() must still return the oracle-correct roll (or raise) - never the roll of the broken launch: the chain
    drains, dr_finish sees the flag, the engine switches itself to per-phase launches and the chain is re-run."""
    m, x, noise, ref = _timeout_fixture()
    eng = m.engine
    eng.set_option("fused_stack", 2)
    good, _ = m.sample(x, None, noise=noise)
    assert eng.fallbacks == 0 and maxdiff(good.cpu(), ref) <= ATOL_STEP
    eng.set_option("stack_fault_test", 1)
    t0 = time.perf_counter()
    roll, _ = m.sample(x, None, noise=noise)                  # ONE call
    assert time.perf_counter() - t0 < 60.0
    assert maxdiff(roll.cpu(), ref) <= ATOL_STEP
    assert eng.fallbacks == 1
    # healed: the engine runs per-phase launches now - more calls just work, also through the other entry points
    again, _ = m.sample(x, None, noise=noise)
    assert torch.equal(again, roll) and eng.fallbacks == 1
    step, _ = m.reverse_diffusion(x, None, 2, noise=noise[2])
    assert bool(torch.isfinite(step).all())
    # and the fused kernel can be switched back on once the device is the engine's own again
    eng.set_option("stack_fault_test", 0)
    eng.set_option("fused_stack", 2)
    n0 = (eng.stack_status(), eng.stack_launches)[1]
    back, _ = m.sample(x, None, noise=noise)
    eng.stack_status()
    assert eng.stack_launches > n0 and eng.fallbacks == 1
    assert maxdiff(back.cpu(), ref) <= ATOL_STEP and maxdiff(back.cpu(), good.cpu()) == 0.0


def test_one_step_and_forward_are_verified_too():
    """The samplers' one-step methods and forward() hand out finished tensors as the reference does: with the fault
    hook on, a single reverse_diffusion() / forward() call returns the right values (healed), never the broken ones."""
    m, x, noise, _ = _timeout_fixture()
    eng = m.engine
    eng.set_option("fused_stack", 0)
    want_step, _ = m.reverse_diffusion(x, None, 2, noise=noise[2])
    wav0 = torch.zeros(8, 64 * 512)
    want_fwd, _ = m(x, wav0, torch.tensor(3).repeat(8), sampling=True)
    for call in ("step", "forward"):
        eng.set_option("fused_stack", 2)
        eng.set_option("stack_fault_test", 1)
        fb = eng.fallbacks
        if call == "step":
            got, _ = m.reverse_diffusion(x, None, 2, noise=noise[2])
            assert maxdiff(got.cpu(), want_step.cpu()) <= 5e-6
        else:
            got, _ = m(x, wav0, torch.tensor(3).repeat(8), sampling=True)
            assert maxdiff(got.cpu(), want_fwd.cpu()) <= 5e-6
        assert eng.fallbacks == fb + 1
        eng.set_option("stack_fault_test", 0)


def test_unchecked_timeout_is_loud_at_the_consume_point_and_heals():
    """The asynchronous form: Engine.sample(check=False) returns at once; finish() is the consume point and raises
    EngineTimeout after a time-out (having healed the engine); until then every other call refuses to start."""
    from diffroll_amd.engine import EngineTimeout
    m, x, noise, ref = _timeout_fixture()
    eng = m.engine
    xb = x.squeeze(1).to(eng.device).contiguous()
    z = noise.reshape(4, 8, 64, 88).to(eng.device).contiguous()
    eng.set_option("fused_stack", 2)
    eng.set_option("stack_fault_test", 1)
    work = xb.clone()
    t0 = time.perf_counter()
    eng.sample("generation_ddpm_x0", work, z, check=False)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 60.0
    with pytest.raises(EngineTimeout):                        # pending and unchecked: nothing else may start
        eng.step("generation_ddpm_x0", xb.clone(), z[2], 2)
    # ... and nothing may CONSUME the invalid roll through the C-ABI either: note extraction, frame counts
    # and the forward-process arithmetic given the engine handle answer DR_ETIMEOUT until dr_finish has been called
    assert eng.pending_timeout()
    with pytest.raises(EngineTimeout):
        eng.note_runs(work, 0.5)
    with pytest.raises(EngineTimeout):
        eng.frame_counts(work, work, 0.5)
    with pytest.raises(EngineTimeout, match="recomputed"):
        eng.finish()
    assert eng.fallbacks == 1
    eng.finish()                                              # cleared: a second check is clean
    assert not eng.pending_timeout()
    work = xb.clone()
    eng.sample("generation_ddpm_x0", work, z, check=False)    # per-phase launches now
    eng.finish()
    assert maxdiff(work.cpu().unsqueeze(1), ref) <= ATOL_STEP
    eng.set_option("stack_fault_test", 0)




def test_gather_reports_an_invalid_shard_collectively():
    """dr_gather with the engine handle while a time-out is pending: the rank still takes part in the collective, its status
    word travels behind the rolls, and the call answers DR_ETIMEOUT naming the rank (with one rank: itself) - on every
    rank, which is what the peers of a timed-out rank need (they used to get the garbage shard with DR_OK)."""
    import ctypes as C
    from diffroll_amd import _cabi
    from diffroll_amd.distributed import NativeComm
    from diffroll_amd.engine import EngineTimeout
    m, x, noise, ref = _timeout_fixture()
    eng = m.engine
    comm = NativeComm(eng.device, rank=0, world_size=1)
    xb = x.squeeze(1).to(eng.device).contiguous()
    z = noise.reshape(4, 8, 64, 88).to(eng.device).contiguous()
    eng.set_option("fused_stack", 2)
    good = xb.clone()
    eng.sample("generation_ddpm_x0", good, z, check=False)
    full = comm.all_gather(good, engine=eng)                  # healthy: DR_OK, the shard comes back
    assert torch.equal(full, good)
    eng.set_option("stack_fault_test", 1)
    work = xb.clone()
    eng.sample("generation_ddpm_x0", work, z, check=False)
    with pytest.raises(RuntimeError, match="rank 0"):
        comm.all_gather(work, engine=eng)
    lib = _cabi.load_library()
    out = torch.empty_like(work)
    rc = lib.dr_gather(eng.h, comm.h, work.data_ptr(), out.data_ptr(), work.shape[0], work.shape[1],
                       torch.cuda.current_stream(eng.device).cuda_stream)
    assert rc == _cabi.DR_ETIMEOUT and b"every rank" in lib.dr_comm_last_error()
    with pytest.raises(EngineTimeout):
        eng.finish()                                          # heals; the shard is recomputed on per-phase launches ...
    eng.set_option("stack_fault_test", 0)
    work = xb.clone()
    eng.sample("generation_ddpm_x0", work, z, check=False)
    full = comm.all_gather(work, engine=eng)                  # ... and gathered again: valid
    assert maxdiff(full.cpu().unsqueeze(1), ref) <= ATOL_STEP
    comm.close()
