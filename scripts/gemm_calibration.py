"""What the vendor library reaches on this part for the GEMMs the hot convolutions are equivalent to (fp16 in, fp32 accumulate, random and
zero operands: the chip clocks to its power budget, so the rate is data-dependent) -- the practical MFMA ceiling the kernels of DESIGN.md
section 4 are to be read against.   python scripts/gemm_calibration.py > profiles/r05_hipblaslt_calibration.txt   (GPU)"""
import torch


def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


# square; tall; the SFT conv1 (825 -> 384, 3x3, N = 4 at 448^2) and the 8x8 stride-4 conv (128 -> 128, N = 4) as plain GEMMs
for (m, n, k) in ((8192, 8192, 8192), (16384, 8192, 4096), (802816, 384, 825 * 9 // 8 * 8), (802816, 128, 8192)):
    for kind in ("randn", "zeros"):
        mk = (lambda *s: torch.randn(*s, device="cuda", dtype=torch.float16)) if kind == "randn" else (lambda *s: torch.zeros(*s, device="cuda", dtype=torch.float16))
        a, b = mk(m, k), mk(k, n)
        ms = t(lambda: torch.matmul(a, b))
        print(f"hipBLASLt fp16 GEMM {m} x {n} x {k} ({kind}): {ms:.3f} ms  {2.0 * m * n * k / ms / 1e9:.0f} TF/s")
        del a, b
