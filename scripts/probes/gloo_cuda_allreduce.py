"""Does a gloo all-reduce of a CUDA tensor complete with two ranks on one GPU?  (rehearsal aid)"""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(60, exit=True)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
r = dist.get_rank()
for n in (1000, 100_000, 920_000, 12_000_000):
    t = torch.full((n,), float(r + 1), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {r}: n={n} sum={t[0].item()} in {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
dist.destroy_process_group()
