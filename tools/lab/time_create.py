import time, torch, sys
sys.path.insert(0, '/root/repo')
import bench
t0=time.perf_counter(); m=bench.build_model(torch.device('cuda',0)); t1=time.perf_counter(); m.engine; torch.cuda.synchronize(); t2=time.perf_counter()
print(f"model build {t1-t0:.1f}s engine create+commit {t2-t1:.1f}s")
