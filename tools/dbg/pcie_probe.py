import time, torch, threading
dev = torch.device("cuda", 0)
n = 600 * 1024 * 1024
h_up = torch.empty(n, dtype=torch.uint8).pin_memory(); h_dn = torch.empty(n, dtype=torch.uint8).pin_memory()
d_up = torch.empty(n, dtype=torch.uint8, device=dev); d_dn = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def up():
    with torch.cuda.stream(s1):
        d_up.copy_(h_up, non_blocking=True)
def dn():
    with torch.cuda.stream(s2):
        h_dn.copy_(d_dn, non_blocking=True)
for name, fns in (("H2D", [up]), ("D2H", [dn]), ("both", [up, dn])):
    for f in fns: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        for f in fns: f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(name, "%.2f ms" % (dt * 1e3), "%.1f GB/s per direction" % (n / dt / 1e9))
# chunked: 6 smaller arrays like the wire batch
parts = [torch.empty(n // 6, dtype=torch.uint8).pin_memory() for _ in range(6)]
dparts = [torch.empty(n // 6, dtype=torch.uint8, device=dev) for _ in range(6)]
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    with torch.cuda.stream(s1):
        for a, b in zip(parts, dparts): b.copy_(a, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("H2D 6 pieces %.2f ms %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
