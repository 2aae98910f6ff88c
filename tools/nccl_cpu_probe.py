"""Diagnostic: how much CPU does one small RCCL gather per iteration cost this process (incl. background threads)?"""
import os, time, sys
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
mode = sys.argv[1] if len(sys.argv) > 1 else "gather"
buf = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
out = [torch.empty_like(buf)]
def cpu(): t = os.times(); return t.user + t.system
for it in range(3):
    dist.gather(buf, out, dst=0); torch.cuda.synchronize()
c0, w0 = cpu(), time.time()
for it in range(40):
    if mode == "gather":
        dist.gather(buf, out, dst=0)
    elif mode == "gather_sync":
        dist.gather(buf, out, dst=0); torch.cuda.synchronize()
    elif mode == "allreduce":
        dist.all_reduce(buf[:1024])
    time.sleep(0.02)
print(mode, "cpu_s", round(cpu() - c0, 2), "wall_s", round(time.time() - w0, 2), flush=True)
dist.destroy_process_group()
