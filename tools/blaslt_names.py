import torch
dev = "cuda:0"
for (M, K, N) in [(25600, 1024, 512), (25600, 768, 512), (102400, 768, 256), (409600, 192, 256), (25600, 512, 512)]:
    a = torch.randn(M, K, device=dev).half(); b = torch.randn(K, N, device=dev).half()
    for _ in range(3):
        c = torch.matmul(a, b)
torch.cuda.synchronize()
