import torch, time
x = torch.empty((4096, 8, 128, 128), dtype=torch.int32, device='cuda')
y = torch.empty_like(x)
for name, f in (('zero_', lambda: x.zero_()), ('fill_(7)', lambda: x.fill_(7)), ('copy_', lambda: y.copy_(x))):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
    print('%s: %.0f us -> %.2f TB/s %s' % (name, us, x.numel() * 4 / us / 1e6, '(read+write: x2)' if name == 'copy_' else ''))
