"""A/B of conv_igemm_glds build variants (CSBSR_LIB=...) on the detector's wide layers, plain and split-fp16 forward.
    python scripts/glds_spread_ab.py                      (parent: runs itself once per library variant)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = [("default", "libcsbsr_hip.so", None), ("variant", "libcsbsr_hip_variant.so", None)]      # (name, library under csbsr_amd/ -- build the variant with e.g. -DGLDS_SPREAD=0 --, CSBSR_CONV_GLDS mode)
SHAPES = [  # name, N, H, W, cin, cout, k, stride, pad, dil
    ("res512 d2", 8, 224, 224, 512, 512, 3, 1, 2, 2), ("res256 d1", 8, 224, 224, 256, 256, 3, 1, 1, 1), ("up_1 1024>256", 8, 448, 448, 1024, 256, 3, 1, 1, 1),
    ("sft825>384", 4, 448, 448, 832, 384, 3, 1, 1, 1), ("bott 2560>1024 1x1", 8, 224, 224, 2560, 1024, 1, 1, 0, 1)]


def child():
    import torch
    from csbsr_amd import _lib as L
    from csbsr_amd.engine import Engine, Conv, FM, pad8
    L.load().csbsr_debug_set_conv_x3(0)
    eng = Engine()
    for name, N, H, W, cin, cout, k, s, p, d in SHAPES:
        w = torch.randn(cout, cin, k, k, device="cuda") / (cin * k * k) ** 0.5
        conv = Conv(eng, "l", {"l.weight": w}, k, s, p, d, bias=False)
        flops = 2.0 * N * H * W * cout * cin * k * k
        out = []
        for mode in ("plain", "split"):
            if mode == "plain":
                x = FM(torch.randn(N, H, W, pad8(cin), device="cuda", dtype=torch.float16), cin)
            else:
                x = eng.new(N, H, W, cin, split=True)
                x.t.normal_()
                x.t.as_strided(x.t.shape, x.t.stride(), x.t.storage_offset() + x.lo).normal_(0, 3e-4)
            y = eng.new(N, H, W, cout, split=(mode == "split"))
            conv.fwd(x, out=y); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8): conv.fwd(x, out=y)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 8
            out.append(f"{mode} {ms:6.3f} ms {flops / ms / 1e9:6.0f} TF/s alg")
        print(f"   {name:22s} " + "   ".join(out), flush=True)


if __name__ == "__main__":
    if os.environ.get("GLDS_AB_CHILD"):
        child()
    else:
        for rep in range(2):
            for vn, lib, mode in VARIANTS:
                path = os.path.join(ROOT, "csbsr_amd", lib)
                if not os.path.exists(path): continue
                print(f"[{vn}] pass {rep}", flush=True)
                env = dict(os.environ, GLDS_AB_CHILD="1", CSBSR_LIB=path)
                if mode: env["CSBSR_CONV_GLDS"] = mode
                subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
