"""Per-call cost of an asynchronous single-rank RCCL all-reduce interleaved with graph replays (what the layer-segmented
DP schedule pays besides the transfer itself)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MTN_FORCE_DIST", "1")
os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("RANK", "0")
import torch, torch.distributed as dist
from mtn_amd import dp
dp.init_distributed()
dev = torch.device("cuda:0")
x = torch.randn(4096, 4096, device=dev)
g = torch.cuda.CUDAGraph()
y = torch.empty_like(x)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3): torch.mm(x, x, out=y)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(8): torch.mm(x, x, out=y)
big = torch.zeros(17_000_000, device=dev)
small = torch.zeros(64, device=dev)
def run(buf, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        works = []
        for k in range(7):
            g.replay()
            if buf is not None: works.append(dist.all_reduce(buf, async_op=True))
        for w in works: w.wait()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for name, buf in (("no exchange", None), ("64 floats", small), ("17M floats", big)):
    run(buf, 5)
    print(f"7 x [graph + all_reduce({name})]: {run(buf):.3f} ms")
dist.destroy_process_group()
