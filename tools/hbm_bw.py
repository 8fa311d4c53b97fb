import torch, time
dev=torch.device('cuda',0)
n=1<<28  # 1 GiB of float32
a=torch.empty(n,device=dev); b=torch.empty(n,device=dev)
def t(fn,it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
ms=t(lambda: a.fill_(1.0)); print('fill  1 GiB: %.3f ms  %.2f TB/s written' % (ms, n*4/ms/1e9))
ms=t(lambda: b.copy_(a)); print('copy  1 GiB: %.3f ms  %.2f TB/s read + %.2f TB/s written' % (ms, n*4/ms/1e9, n*4/ms/1e9))
ms=t(lambda: a.sum()); print('sum   1 GiB: %.3f ms  %.2f TB/s read' % (ms, n*4/ms/1e9))
ms=t(lambda: torch.add(a,b,out=b)); print('add   out=b: %.3f ms  %.2f TB/s total' % (ms, 3*n*4/ms/1e9))
