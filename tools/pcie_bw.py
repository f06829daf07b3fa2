import torch, time
for mb in (0.465, 8, 64, 256):
    n=int(mb*1e6)
    h=torch.empty(n,dtype=torch.uint8).pin_memory(); d=torch.empty(n,dtype=torch.uint8,device='cuda')
    for _ in range(3): d.copy_(h,non_blocking=True)
    torch.cuda.synchronize(); reps=max(4,int(512/mb)); t=time.perf_counter()
    for _ in range(reps): d.copy_(h,non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("H2D %8.3f MB x %d: %.1f GB/s" % (mb,reps,mb*reps/1e3/dt))
    t=time.perf_counter()
    for _ in range(reps): h.copy_(d,non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    print("D2H %8.3f MB x %d: %.1f GB/s" % (mb,reps,mb*reps/1e3/dt))
