"""Chip copy rate for the compaction's shape: 5.3 GB int32 -> 5.3 GB, aligned and dword-misaligned source."""
import torch, time
n = 1336366087
src = torch.empty(n + 16, dtype=torch.int32, device="cuda"); src.random_(0, 1 << 22)
dst = torch.empty(n, dtype=torch.int32, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for off in (0, 1, 3):
    ms = t(lambda: dst.copy_(src[off:off + n]))
    print("copy_ offset %d dwords: %.3f ms  %.2f TB/s (read+write)" % (off, ms, 2 * 4 * n / ms / 1e9))
ms = t(lambda: dst.zero_())
print("fill: %.3f ms  %.2f TB/s" % (ms, 4 * n / ms / 1e9))
ms = t(lambda: src.sum())
print("read (sum): %.3f ms  %.2f TB/s" % (ms, 4 * n / ms / 1e9))
