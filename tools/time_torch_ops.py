import time, torch
x = torch.zeros(1024, dtype=torch.long, device='cuda'); f = (torch.rand(1024, device='cuda') > 0.5).to(torch.uint8); d = torch.zeros(1024, dtype=torch.uint8, device='cuda')
A = torch.rand(80, 1024, 1, 4, device='cuda'); ar = torch.arange(1024, device='cuda')
def t(name, fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('%-28s queue %.3f ms/op, with final sync %.3f ms/op' % (name, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
t('x += 1', lambda: x.add_(1))
t('live = f * (1 - d)', lambda: f * (1 - d))
t('to(long)', lambda: f.to(torch.long))
t('A[x.clamp(max=79), ar]', lambda: A[x.clamp(max=79), ar])
t('int(f.sum())', lambda: int(f.sum()), 50)
