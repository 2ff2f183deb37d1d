"""How exact is torch's fp32 matmul (rocBLAS / hipBLASLt) on this box?  GCN-sized GEMMs vs float64 on the CPU."""
import torch
torch.manual_seed(0)
for (m, k, n) in [(2304, 768, 512), (2304, 512, 1280), (288, 512, 512), (72, 768, 512), (512, 2304, 768)]:
    a, b = torch.randn(m, k), torch.randn(k, n)
    ref = (a.double() @ b.double())
    cpu = (a @ b).double()
    gpu = (a.cuda() @ b.cuda()).cpu().double()
    lin = torch.nn.functional.linear(a.cuda(), b.t().contiguous().cuda()).cpu().double()
    sc = float(ref.abs().max())
    print(f"{m}x{k}x{n}: cpu {float((cpu - ref).abs().max()) / sc:.2e}  gpu mm {float((gpu - ref).abs().max()) / sc:.2e}  "
          f"gpu linear {float((lin - ref).abs().max()) / sc:.2e}")
print("allow_tf32", torch.backends.cuda.matmul.allow_tf32, "preferred blas", torch.backends.cuda.preferred_blas_library())
