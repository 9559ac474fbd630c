"""Developer script: torch.distributed all_to_all_single with explicit split sizes and async_op on the NCCL (= RCCL) backend,
world size 1 (all a one-GPU box can do): the call shape RowBlockExchange.step uses."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
send = torch.arange(4096 * 16, dtype=torch.float32, device="cuda")
recv = torch.zeros(4096 * 20, dtype=torch.float32, device="cuda")
w = dist.all_to_all_single(recv[:4096 * 16], send, [4096 * 16], [4096 * 16], async_op=True)
x = torch.rand(1 << 20, device="cuda").sum()          # work on the compute stream meanwhile
w.wait()
torch.cuda.synchronize()
assert torch.equal(recv[:4096 * 16], send)
w = dist.all_to_all_single(recv[:0], send[:0], [0], [0], async_op=True)     # a rank with nothing to exchange still makes the call
w.wait()
objs = [None]
dist.all_gather_object(objs, [[1, 2], []])
assert objs == [[[1, 2], []]]
dist.destroy_process_group()
print("all_to_all_single(split sizes, async) on nccl: ok")
