import torch, time
torch.cuda.init(); torch.cuda.synchronize()
for gb in (1, 16, 69):
    t = time.time(); x = torch.empty(int(gb * (1 << 30)), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t1 = time.time() - t
    t = time.time(); x.zero_(); torch.cuda.synchronize(); t2 = time.time() - t
    t = time.time(); x.zero_(); torch.cuda.synchronize(); t3 = time.time() - t
    del x; torch.cuda.empty_cache()
    print("%d GiB: alloc %.3f s, first touch (memset) %.3f s, second memset %.3f s" % (gb, t1, t2, t3))
