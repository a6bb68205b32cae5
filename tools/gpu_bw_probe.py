"""What does this box sustain? torch copy / reduction bandwidth on the same buffers the bench uses."""
import torch, time
x = torch.randint(0, 255, (4, 8192, 8192), dtype=torch.int32, device="cuda")
y = torch.empty_like(x)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
ms = t(lambda: y.copy_(x)); print(f"copy 1.07GB: {ms:.3f} ms -> {2*x.numel()*4/ms/1e9:.2f} TB/s (r+w)")
ms = t(lambda: y[3].copy_(x[3])); print(f"copy 268MB: {ms:.3f} ms -> {2*x[3].numel()*4/ms/1e9:.2f} TB/s (r+w)")
ms = t(lambda: x[3].max()); print(f"max-reduce 268MB: {ms:.3f} ms -> {x[3].numel()*4/ms/1e9:.2f} TB/s (read)")
ms = t(lambda: x[:3].max()); print(f"max-reduce 805MB: {ms:.3f} ms -> {x[:3].numel()*4/ms/1e9:.2f} TB/s (read)")
ms = t(lambda: y.zero_()); print(f"memset 1.07GB: {ms:.3f} ms -> {x.numel()*4/ms/1e9:.2f} TB/s (write)")
