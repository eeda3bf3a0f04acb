import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
torch.cuda.set_device(0)
x = torch.ones(127_000_000 // 4, device="cuda")
for n in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    dist.all_reduce(x); torch.cuda.synchronize()
    if dist.get_rank() == 0: print(f"gloo all_reduce of {x.numel() * 4 / 1e6:.0f} MB: {(time.perf_counter() - t0) * 1e3:.0f} ms")
dist.destroy_process_group()
