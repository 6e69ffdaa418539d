# dev: what the vendor library's fp32 GEMM reaches on the default geometry's gate contraction (M = rows of a chunk, N = 2C, K = Ktp) and on the
# paper-size weight-gradient shape -- the yardstick for k_gemm_nn / k_gemm_tn / k_wgrad3 (torch.matmul -> hipBLASLt / rocBLAS, TF32 off)
import time, torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
def rate(M, N, K, tn=False, iters=30):
    a = torch.randn((K, M) if tn else (M, K), device=dev); b = torch.randn(K, N, device=dev)
    f = (lambda: a.t() @ b) if tn else (lambda: a @ b)
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    return 2.0 * M * N * K / dt / 1e12, dt * 1e6
for name, M, N, K, tn in (("default gate   [20299 x 1072] . [1072 x 1024]", 20299, 1024, 1072, False),
                          ("default dZ.W1^T [20299 x 1024] . [1024 x 1072]", 20299, 1072, 1024, False),
                          ("default dW1    [1024 x 20299] . [20299 x 1072] (time contraction)", 1024, 1072, 20299, True),
                          ("paper post-net [20000 x 256] . [256 x 256]", 20000, 256, 256, False),
                          ("paper dW1      [128 x 20299] . [20299 x 176] (time contraction)", 128, 176, 20299, True)):
    tf, us = rate(M, N, K, tn)
    print("%-70s %7.1f TFLOP/s  %8.1f us  (%.2f of 157.3)" % (name, tf, us, tf / 157.3))
