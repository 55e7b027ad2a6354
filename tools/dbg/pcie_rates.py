"""Raw rates on this box: DMA device <-> pinned host memory, the runtime's pageable path, one thread's memcpy."""
import time, numpy as np, torch
dev = torch.device("cuda:0")
for mb in (0.5, 1.5, 13.0):
    n = int(mb * 1e6) // 4
    d = torch.arange(n, dtype=torch.float32, device=dev)
    hp = torch.empty(n, dtype=torch.float32).pin_memory()
    hq = torch.empty(n, dtype=torch.float32)
    hq2 = np.empty(n, np.float32)
    def timed(fn, reps=30):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6
    a = timed(lambda: (hp.copy_(d, non_blocking=True), torch.cuda.synchronize()))
    b = timed(lambda: hq.copy_(d))
    c = timed(lambda: (d.copy_(hp, non_blocking=True), torch.cuda.synchronize()))
    e = timed(lambda: d.copy_(hq))
    src = hp.numpy()
    f = timed(lambda: np.copyto(hq2, src))
    print(f"{mb:5.1f} MB: D2H pinned {a:7.1f} us ({mb*1e3/a:5.1f} GB/s)  D2H pageable {b:7.1f} us ({mb*1e3/b:5.1f})  H2D pinned {c:7.1f} us ({mb*1e3/c:5.1f})  "
          f"H2D pageable {e:7.1f} us ({mb*1e3/e:5.1f})  memcpy pinned->pageable one thread {f:7.1f} us ({mb*1e3/f:5.1f})", flush=True)
