import os, torch, torch.distributed as dist
kw = {}
dist.init_process_group("cpu:gloo,cuda:nccl", **kw)
r = dist.get_rank()
t = torch.ones(1, dtype=torch.int32) * (r + 1)
dist.all_reduce(t, op=dist.ReduceOp.MIN)
box = [b"x" * 128 if r == 0 else None]
dist.broadcast_object_list(box, src=0)
tt = torch.tensor([float(r)], dtype=torch.float64)
dist.all_reduce(tt, op=dist.ReduceOp.MAX)
lst = [torch.zeros(1, dtype=torch.float64) for _ in range(dist.get_world_size())]
dist.all_gather(lst, tt)
print("rank", r, "min", int(t.item()), "box", len(box[0]), "max", float(tt.item()), [float(v) for v in lst], flush=True)
dist.destroy_process_group()
