"""PCIe copy rates of this box between pinned host memory and HBM: one stream, two streams, both directions at once
(what bounds the whole-stream API: DESIGN.md section 5)"""
import time, torch
dev = torch.device("cuda", 0)
N = 512 << 20
h = [torch.empty(N, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
d = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(2)]
s = [torch.cuda.Stream(dev) for _ in range(2)]
def run(name, jobs, reps=5):
    for r in range(reps + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k, (dst, src) in enumerate(jobs):
            with torch.cuda.stream(s[k % 2]):
                dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name:48s} {sum(x[0].numel() for x in jobs) / dt / 1e9:7.1f} GB/s ({dt * 1e3:.1f} ms)", flush=True)
run("D2H one stream, 512 MiB", [(h[0], d[0])])
run("H2D one stream, 512 MiB", [(d[0], h[0])])
run("D2H two streams, 2 x 256 MiB", [(h[0][:N // 2], d[0][:N // 2]), (h[1][:N // 2], d[1][:N // 2])])
run("H2D two streams, 2 x 256 MiB", [(d[0][:N // 2], h[0][:N // 2]), (d[1][:N // 2], h[1][:N // 2])])
run("H2D + D2H at once, 512 MiB each", [(d[0], h[0]), (h[1], d[1])])
for mb in (1, 4, 16, 64):
    n = mb << 20
    run(f"D2H one stream, {mb} MiB", [(h[0][:n], d[0][:n])])
