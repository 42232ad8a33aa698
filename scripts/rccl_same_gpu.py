"""experiment: can two RCCL ranks share one GPU on this stack? (decides how the N>1 exchange can be rehearsed on a 1-GPU box)"""
import os, sys, datetime
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=datetime.timedelta(seconds=40))
    x = torch.arange(4 * world, dtype=torch.int32, device=dev) + 100 * rank
    y = torch.empty_like(x)
    dist.all_to_all_single(y, x)
    torch.cuda.synchronize()
    print("rank", rank, "all_to_all ok", y.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", rank, "FAILED:", type(e).__name__, str(e)[:400], flush=True)
    os._exit(1)
