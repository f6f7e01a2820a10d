"""Experiment: fp32 GEMM vs three bf16 GEMMs with fp32 output (hi/lo split) through hipBLASLt on MI355X."""
import torch, time
def t(fn, n=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def split(a):
    hi = a.bfloat16(); lo = (a - hi.float()).bfloat16(); return hi, lo
for (M, K, N) in [(65536, 1024, 256), (65536, 256, 1024), (65536, 2048, 512), (65536, 512, 2048), (65536, 256, 256)]:
    x = torch.randn(M, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05
    ref = (x.double() @ w.double())
    f32 = lambda: torch.mm(x, w)
    xh, xl = split(x); wh, wl = split(w)
    def s3():
        y = torch.mm(xh, wh, out_dtype=torch.float32)
        y = torch.addmm(y, xh, wl, out_dtype=torch.float32)
        y = torch.addmm(y, xl, wh, out_dtype=torch.float32)
        return y
    def s3_split():
        a, b = split(x); return s3()
    b16 = lambda: torch.mm(xh, wh, out_dtype=torch.float32)
    try:
        e32 = ((f32().double() - ref).abs().max() / ref.abs().max()).item()
        e3 = ((s3().double() - ref).abs().max() / ref.abs().max()).item()
        e1 = ((b16().double() - ref).abs().max() / ref.abs().max()).item()
        fl = 2.0 * M * K * N / 1e9
        a, b, c, d = t(f32), t(b16), t(s3), t(lambda: split(x))
        print("M%d K%d N%d: fp32 %.3f ms (%.0f TF) | bf16->f32 %.3f ms (%.0f TF) | split3 %.3f ms | split op %.3f ms | err fp32 %.1e bf16 %.1e split3 %.1e"
              % (M, K, N, a, fl / a, b, fl / b, c, d, e32, e1, e3), flush=True)
    except Exception as e:
        print("failed", (M, K, N), repr(e)[:300])
