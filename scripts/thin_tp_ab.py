"""kb.up_conv1 forward (3 -> 128 transposed 8x8 s4 + PReLU + residual add, csrc/conv_thin.hip conv_thin_tp_kernel): ms per launch at N = 4,
LR 448 -> HR 1792, per library variant (CSBSR_LIB).   python scripts/thin_tp_ab.py   (GPU)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("TP_AB_CHILD"):
    import torch
    from csbsr_amd import _lib as L
    from csbsr_amd.engine import Engine, Conv, FM, pad8
    eng = Engine()
    params = {"l.weight": torch.randn(3, 128, 8, 8, device="cuda") / 14.0, "l.a": torch.full((1,), 0.25, device="cuda")}
    conv = Conv(eng, "l", params, 8, 4, 2, 1, transposed=True, bias=False, act=L.ACT_PRELU, prelu="l.a")
    x = FM(torch.randn(4, 448, 448, 8, device="cuda", dtype=torch.float16), 3)
    x.t[..., 3:] = 0
    res = FM(torch.randn(4, 1792, 1792, 128, device="cuda", dtype=torch.float16), 128)
    out = eng.new(4, 1792, 1792, 128)
    conv.fwd(x, out=out, res=res, res_mode=L.RES_ADD); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): conv.fwd(x, out=out, res=res, res_mode=L.RES_ADD)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{os.path.basename(os.environ.get('CSBSR_LIB', 'default')):28s} kernel {int(L.load().csbsr_debug_last_conv_kernel())}: {ms:.3f} ms  {2 * 4 * 1792 * 1792 * 256 / ms / 1e9:.2f} TB/s  checksum {float(out.t.float().abs().sum()):.6e}")
else:
    for lib in ("libcsbsr_hip_variant.so", "libcsbsr_hip.so", "libcsbsr_hip_variant.so", "libcsbsr_hip.so"):
        path = os.path.join(ROOT, "csbsr_amd", lib)
        if os.path.exists(path):
            subprocess.run([sys.executable, __file__], env=dict(os.environ, TP_AB_CHILD="1", CSBSR_LIB=path))
