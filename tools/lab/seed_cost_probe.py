import sys, time, torch
sys.path.insert(0, '/root/repo')
from tools import tuning_env  # noqa: E402

tuning_env.install()        # DR_TEST_TUNE="tune.stack_fl=2,..." pins engine options for this process
import bench
dev = torch.device('cuda', 0)
m = bench.build_model(dev)
g = torch.Generator().manual_seed(0)
wav = (0.1 * torch.randn(16, 64000, generator=g)).to(dev)
x = torch.randn(16, 1, 125, 88, generator=g).to(dev)
m.sample(x, wav, seed=0); torch.cuda.synchronize()
for seed in (0, 0, 1, 2, 2, 3):
    t0 = time.perf_counter(); m.sample(x, wav, seed=seed); torch.cuda.synchronize()
    print(f"seed {seed}: {1e3 * (time.perf_counter() - t0):.1f} ms")
