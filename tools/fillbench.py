import torch, time
x = torch.empty(65536, 263, device='cuda'); y = torch.empty_like(x)
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n*1e3
print('fill 69MB: %.1f us' % t(lambda: x.fill_(1.0)))
print('copy 69MB->69MB: %.1f us' % t(lambda: y.copy_(x)))
print('x*2 in place: %.1f us' % t(lambda: x.mul_(2.0)))
