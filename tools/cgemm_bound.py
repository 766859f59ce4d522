import torch, time
torch.manual_seed(0)
F = 3672
for B in (64, 256):
    A = torch.randn(F, B, 512, dtype=torch.complex64, device='cuda')
    W = torch.randn(F, 512, 512, dtype=torch.complex64, device='cuda')
    for _ in range(2): Y = torch.bmm(A, W)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): Y = torch.bmm(A, W)
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    fl = F * B * 512 * 512 * 8
    print('complex64 bmm B=%d: %.2f ms  %.1f TFLOP/s (real flops)' % (B, dt * 1e3, fl / dt / 1e12))
    # real formulation: [Ar Ai] x [[Wr Wi],[-Wi Wr]]
    Ar = torch.randn(F, B, 1024, device='cuda'); Wr = torch.randn(F, 1024, 1024, device='cuda')
    for _ in range(2): Y2 = torch.bmm(Ar, Wr)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): Y2 = torch.bmm(Ar, Wr)
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    print('fp32 real bmm  B=%d: %.2f ms  %.1f TFLOP/s' % (B, dt * 1e3, F * B * 1024 * 1024 * 2 / dt / 1e12))
    del A, W, Y, Ar, Wr, Y2
