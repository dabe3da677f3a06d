import torch, time
n = 256_900_000
a = torch.empty(n, dtype=torch.int64, device="cuda").random_()
b = torch.empty_like(a)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: b.copy_(a)); print("copy 2x%.2f GB: %.3f ms  %.2f TB/s (r+w)" % (n*8/1e9, ms, 2*n*8/ms/1e9))
ms = t(lambda: a.sum()); print("sum read %.2f GB: %.3f ms  %.2f TB/s" % (n*8/1e9, ms, n*8/ms/1e9))
ms = t(lambda: b.fill_(1)); print("fill write %.2f GB: %.3f ms  %.2f TB/s" % (n*8/1e9, ms, n*8/ms/1e9))
a32 = a.view(torch.int32); 
ms = t(lambda: torch.add(a, 1, out=b)); print("add r+w: %.3f ms  %.2f TB/s" % (ms, 2*n*8/ms/1e9))
