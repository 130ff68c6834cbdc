import torch, time
dev=torch.device("cuda:0")
x=torch.empty(256*1024*1024//8*4, dtype=torch.float64, device=dev).normal_()   # 1 GiB
y=torch.empty_like(x)
def t(fn,n=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e-3
gb=x.numel()*8/1e9
dt=t(lambda: y.copy_(x)); print("copy  1 GiB: %.1f us  read+write %.2f TB/s" % (dt*1e6, 2*gb/dt/1e3))
dt=t(lambda: x.sum()); print("sum   1 GiB: %.1f us  read %.2f TB/s" % (dt*1e6, gb/dt/1e3))
dt=t(lambda: y.fill_(1.0)); print("fill  1 GiB: %.1f us  write %.2f TB/s" % (dt*1e6, gb/dt/1e3))
xs=x[:32*1024*1024]  # 256 MB
ys=y[:32*1024*1024]
dt=t(lambda: xs.sum()); print("sum 256 MB: %.1f us  read %.2f TB/s" % (dt*1e6, 0.268/dt/1e3))
dt=t(lambda: ys.copy_(xs)); print("copy 256 MB: %.1f us  r+w %.2f TB/s" % (dt*1e6, 2*0.268/dt/1e3))
