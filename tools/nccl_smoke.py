"""Dev-only: the collectives bench.py and shard.py use for N > 1 (barrier, all_reduce MAX on a float64 scalar, all_gather_object,
all_gather of two words, broadcast into a slice, destroy) on the RCCL
backend with the ranks this box has GPUs for (one GPU: world_size 1):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/nccl_smoke.py"""
import os
import torch
import torch.distributed as dist
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
dist.barrier()
t = torch.tensor([1.5 + rank], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
recs = [None] * world
dist.all_gather_object(recs, {"rank": rank, "device": torch.cuda.get_device_name()})      # bench.py: gather_rank_records
sizes = [torch.zeros(2, dtype=torch.int64, device="cuda") for _ in range(world)]
dist.all_gather(sizes, torch.tensor([3 + rank, 7], dtype=torch.int64, device="cuda"))      # shard.replicate_rows: the (n_r, X_r) pairs
piece = torch.arange(8, dtype=torch.int32, device="cuda")
dist.broadcast(piece[2:6], src=0)                                                           # ... and a slice written in place
torch.cuda.synchronize()
print("nccl ok", rank, world, float(t.item()), recs, [s_.tolist() for s_ in sizes], flush=True)
dist.barrier()
dist.destroy_process_group()
