import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _pkg, torch, torch.distributed as dist
pkg = _pkg.load()
from importlib import import_module
sharded = import_module("stellar_random_walk_amd.distributed")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
sw = sharded.ShardedWalker(device=0)
sw.generate_rmat(22, 16 << 22)
sw.walk_iteration(iteration=0, walk_length=80)
torch.cuda.synchronize(); t = time.perf_counter()
st = sw.walk_iteration(iteration=1, walk_length=80)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("iteration %.1f ms, kernel sum %.1f ms, steps %d -> %.2f Gsteps/s" % (dt * 1e3, st["kernel_ms"], st["n_steps"], st["n_steps"] / dt / 1e9))
for nb in (4, 10):
    torch.cuda.synchronize(); t = time.perf_counter()
    st = sw.walk_iteration(iteration=3, walk_length=80, num_walks=nb)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print("%d iterations batched: %.1f ms, kernel sum %.1f ms, steps %d -> %.2f Gsteps/s" % (nb, dt * 1e3, st["kernel_ms"], st["n_steps"], st["n_steps"] / dt / 1e9))
sys.exit(0) if os.environ.get("NOPROF") else None
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); sw.walk_iteration(iteration=2, walk_length=80); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
dist.destroy_process_group()
