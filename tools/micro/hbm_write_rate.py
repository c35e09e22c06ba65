import torch, time
n = 4_800_000_000 // 4
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ms = t(lambda: x.fill_(1.0)); print("fill 4.8 GB: %.3f ms = %.0f GB/s written" % (ms, 4.8e9 / ms / 1e6))
ms = t(lambda: x.zero_()); print("zero (memset) 4.8 GB: %.3f ms = %.0f GB/s written" % (ms, 4.8e9 / ms / 1e6))
ms = t(lambda: y.copy_(x)); print("copy 4.8 GB: %.3f ms = %.0f GB/s read + the same written" % (ms, 4.8e9 / ms / 1e6))
