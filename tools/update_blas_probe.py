"""Time the update-phase GEMM shapes (2M x 64 x 64 forward / backward) under the BLAS back ends
PyTorch-ROCm offers; informs the note on the update phase in DESIGN.md."""
import sys, time, torch
lib = sys.argv[1] if len(sys.argv) > 1 else "default"
if lib != "default":
    torch.backends.cuda.preferred_blas_library(lib)
B = 1 << 21
x = torch.randn(B, 64, device="cuda"); w = torch.randn(64, 64, device="cuda", requires_grad=True); b = torch.zeros(64, device="cuda", requires_grad=True)
lin = lambda: torch.nn.functional.linear(x, w, b)
for _ in range(3):
    y = lin(); y.sum().backward()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    y = lin()
torch.cuda.synchronize(); t1 = time.perf_counter()
for _ in range(10):
    y = lin(); y.backward(torch.ones_like(y))
torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"{lib}: forward {1e2*(t1-t0):.2f} ms, forward+backward {1e2*(t2-t1):.2f} ms per linear(2M x 64 -> 64)")
