import os, sys, time, json, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import fhip_amd as fhe
ctx = fhe.SEALContext.preset("P4096")
blocks, wave = 256, 32
rec = fhe.server.RECORD_HEADER + 2*ctx.k*ctx.n*8
fin, fout = "/dev/shm/dbg_in.ct", "/dev/shm/dbg_out.ct"
host = torch.empty((wave,3,64,2,ctx.k,ctx.n), dtype=torch.int64).pin_memory()
w = fhe.server.StreamFile(fin, write=True, size=blocks*192*rec)
for s in range(0, blocks, wave):
    host.copy_(ctx.random_ct(wave,3,64, seed=1, first_index=s))
    w.transfer(s*192, wave*192, 2, ctx, host, 8)
w.close()
# raw stage rates
r = fhe.server.StreamFile(fin)
for th in (8, 16, 32):
    t0=time.time(); 
    for s in range(0, blocks, wave): r.transfer(s*192, wave*192, 2, ctx, host, th)
    dt=time.time()-t0; print("read threads", th, blocks*192*rec/dt/1e9, "GB/s")
r.close()
o = fhe.server.StreamFile(fout, write=True, size=blocks*192*rec)
for rep in range(3):
  for th in (8, 16):
    t0=time.time()
    for s in range(0, blocks, wave): o.transfer(s*192, wave*192, 2, ctx, host, th)
    dt=time.time()-t0; print("write pass", rep, "threads", th, blocks*192*rec/dt/1e9, "GB/s")
o.close()
dev = torch.empty_like(host, device="cuda")
torch.cuda.synchronize()
t0=time.time()
for _ in range(8): dev.copy_(host, non_blocking=True)
torch.cuda.synchronize(); dt=time.time()-t0; print("H2D", 8*host.numel()*8/dt/1e9, "GB/s")
t0=time.time()
for _ in range(8): host.copy_(dev, non_blocking=True)
torch.cuda.synchronize(); dt=time.time()-t0; print("D2H", 8*host.numel()*8/dt/1e9, "GB/s")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
host2 = torch.empty_like(host).pin_memory(); dev2 = torch.empty_like(dev)
torch.cuda.synchronize(); t0=time.time()
for _ in range(8):
    with torch.cuda.stream(s1): dev.copy_(host, non_blocking=True)
    with torch.cuda.stream(s2): host2.copy_(dev2, non_blocking=True)
torch.cuda.synchronize(); dt=time.time()-t0; print("H2D+D2H concurrent", 16*host.numel()*8/dt/1e9, "GB/s total")
os.remove(fin); os.remove(fout)
