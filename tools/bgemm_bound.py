import torch, time
F = 3672
for B in (256,):
    A = torch.randn(F, B, 1536, dtype=torch.bfloat16, device='cuda')
    W = torch.randn(F, 1536, 512, dtype=torch.bfloat16, device='cuda')
    for _ in range(2): Y = torch.bmm(A, W)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): Y = torch.bmm(A, W)
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    print('bf16 bmm B=%d K=1536: %.2f ms  %.0f TFLOP/s' % (B, dt * 1e3, F * B * 1536 * 512 * 2 / dt / 1e12))
    A = torch.randn(F, B, 512, dtype=torch.bfloat16, device='cuda')
    W = torch.randn(F, 512, 512, dtype=torch.bfloat16, device='cuda')
    for _ in range(2): Y = torch.bmm(A, W)
    torch.cuda.synchronize(); t = time.time()
    for _ in range(3): Y = torch.bmm(A, W)
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    print('bf16 bmm B=%d K=512: %.2f ms  %.0f TFLOP/s' % (B, dt * 1e3, F * B * 512 * 512 * 2 / dt / 1e12))
