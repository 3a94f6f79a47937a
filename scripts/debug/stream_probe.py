"""Which HIP streams run concurrently?  torch.cuda._sleep (a one-wave spin) on several streams at once: streams that share a
hardware queue take the sum of the sleeps, streams on queues of their own the maximum.  With / without RCCL initialised."""
import os, sys, time
import torch

if len(sys.argv) > 1 and sys.argv[1] == "rccl":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(8, device="cuda"); dist.all_reduce(t); dist.barrier()
torch.cuda.set_device(0)
main = torch.cuda.current_stream()
CYC = 2_000_000


def elapsed(streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(CYC)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


elapsed([main]); one = min(elapsed([main]) for _ in range(3))
print("one sleep: %.3f ms" % one)
cands = [torch.cuda.Stream() for _ in range(10)]
for i, c in enumerate(cands):
    print("cand %d with main: %.2f x" % (i, elapsed([main, c]) / one))
print("pairs:")
for i in range(len(cands)):
    print(" ".join("%.1f" % (elapsed([cands[i], cands[j]]) / one) if j != i else " - " for j in range(len(cands))))
