"""Does RCCL accept two ranks on ONE device?  (round-2 question from VERDICT r1 item 1).  Run under torch.distributed.run
with 2 procs; prints the outcome instead of hanging (timeout around it)."""
import os, sys, torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda:0") * (rank + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f"rank {rank}: nccl on a shared device OK -> {t.tolist()}", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: nccl on a shared device REFUSED: {type(e).__name__}: {str(e)[:300]}", flush=True)
    sys.exit(3)
