"""One name per rocprofv3 kernel row (= template instance) of the MFMA kernels, shared by bench.py's roofline block (which knows a launch by
the id csbsr_debug_last_conv_kernel / csbsr_debug_last_wgrad_kernel report, csrc/csbsr_debug.h) and scripts/summarise_profiles.py (which
knows it by the mangled or demangled name in the rocprofv3 CSVs) -- so that "the dominant kernel" is the same row in both."""
import re

_CONV = {0: "conv_igemm_kernel<32,4,1>", 1: "conv_igemm_kernel<64,2,2>", 2: "conv_igemm_kernel<128,2,2>", 5: "conv_thin_cout_kernel",
         6: "conv_thin_cin_kernel", 11: "conv_thin_tp_kernel", 13: "conv_thin_cin2_kernel", 15: "conv_thin_sc_kernel", 16: "conv_thin_tpd_kernel",
         10: "conv_x3_kernel<3>", 12: "conv_x3_kernel<2>", 17: "conv_x3_kernel<3,1024>", 18: "conv_x3_kernel<2,1024>"}
_GLDS = {3: (128, 2, 2, 1), 4: (256, 4, 3, 1), 7: (256, 4, 2, 2), 14: (128, 2, 2, 0)}
_WGRAD = {0: "conv_wgrad_kernel<128,128,2,2>", 1: "conv_wgrad_kernel<128,256,2,4>", 2: "conv_wgrad_kernel<64,128,2,2>",
          3: "conv_wgrad_kernel<32,128,1,4>", 4: "conv_wgrad_thin_kernel", 5: "conv_wgrad_glds_kernel<128,128>",
          6: "conv_wgrad_glds_kernel<128,256>", 7: "conv_wgrad_glds_kernel<256,256>", 8: "conv_wgrad_hr_kernel",
          9: "conv_wgrad_glds_kernel<128,512>"}


def conv_row(kid):
    """csbsr_debug_last_conv_kernel() value (low byte: kernel, bits 8..: template instance) -> row name"""
    base, var = kid & 255, kid >> 8
    if base in _GLDS:
        bm, nwm, ns, ct = _GLDS[base]
        return f"conv_igemm_glds_kernel<{bm},{nwm},{ns},{ct},FS={var & 1},GK={(var >> 1) & 1}>"
    if base == 9:
        return f"conv_tp_kernel<res={var & 1},acc={(var >> 1) & 1},mask={(var >> 2) & 1},sums={(var >> 3) & 1}>"
    if base == 8:
        return (f"conv_hr_kernel<{7 if var & 1 else 4},stat={(var >> 4) & 1},taps={1 if var & 2 else 9},nct={2 if var & 4 else 1},"
                f"mask={(var >> 3) & 1},cb={(var >> 5) & 1}>")
    if base == 10 and var & 1:
        return "conv_x3_kernel<3,2048>"
    if base == 19:
        return f"conv_x3w_kernel<{var}>"
    if base in (21, 22):
        return "head1_fwd_kernel" if base == 21 else "head1_bwd_input_kernel"
    if base == 20:
        return "conv_x3n_kernel<%d,%d%s%s>" % (var & 1, (var >> 1) & 1, ",wide" if var & 4 else "", ",lean" if var & 8 else "")
    return _CONV.get(base, f"conv?{base}")


def wgrad_row(kid):
    return _WGRAD.get(kid, f"conv_wgrad?{kid}")


def family(row):
    """source family of a row: the kernel name without its template arguments"""
    return row.split("<", 1)[0]


def _b(x):
    return "1" if x in ("1", "true") else "0"


def canon(name):
    """rocprofv3 kernel name (mangled _Z..., or demangled with template arguments) -> the row name conv_row / wgrad_row give it"""
    m = (re.search(r"conv_igemm_glds_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)E", name)
         or re.search(r"conv_igemm_glds_kernel<(\d+), (\d+), (\d+), (\d+), (\w+), (\w+)>", name))
    if m:
        g = m.groups()
        return "conv_igemm_glds_kernel<%s,%s,%s,%s,FS=%s,GK=%s>" % (g[0], g[1], g[2], g[3], _b(g[4]), _b(g[5]))
    m = re.search(r"conv_igemm_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name) or re.search(r"conv_igemm_kernel<(\d+), (\d+), (\d+)>", name)
    if m:
        return "conv_igemm_kernel<%s,%s,%s>" % m.groups()
    m = re.search(r"conv_wgrad_glds_kernelILi(\d+)ELi(\d+)E", name) or re.search(r"conv_wgrad_glds_kernel<(\d+), (\d+)", name)
    if m:
        return "conv_wgrad_glds_kernel<%s,%s>" % m.groups()
    m = re.search(r"conv_wgrad_kernel<(\d+), (\d+), (\d+), (\d+)", name) or re.search(r"conv_wgrad_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
    if m:
        return "conv_wgrad_kernel<%s,%s,%s,%s>" % m.groups()
    m = (re.search(r"conv_hr_kernelILi(\d+)ELb(\d)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)E", name)
         or re.search(r"conv_hr_kernel<(\d+), (\w+), (\d+), (\d+), (\w+), (\w+)>", name))
    if m:
        g = m.groups()
        return "conv_hr_kernel<%s,stat=%s,taps=%s,nct=%s,mask=%s,cb=%s>" % (g[0], _b(g[1]), g[2], g[3], _b(g[4]), _b(g[5]))
    m = re.search(r"conv_tp_kernelILi\d+ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name) or re.search(r"conv_tp_kernel<\d+, (\w+), (\w+), (\w+), (\w+)>", name)
    if m:
        return "conv_tp_kernel<res=%s,acc=%s,mask=%s,sums=%s>" % tuple(_b(x) for x in m.groups())
    m = (re.search(r"conv_x3n_kernelILb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
         or re.search(r"conv_x3n_kernel<(\w+), (\w+), (\w+), (\w+)>", name))
    if m:
        return "conv_x3n_kernel<%s,%s%s%s>" % (_b(m.group(1)), _b(m.group(2)), ",wide" if _b(m.group(3)) == "1" else "",
                                               ",lean" if _b(m.group(4)) == "1" else "")
    m = re.search(r"conv_x3w_kernelILi(\d)E", name) or re.search(r"conv_x3w_kernel<(\d)>", name)
    if m:
        return "conv_x3w_kernel<%s>" % m.group(1)
    m = re.search(r"conv_x3_kernelILi(\d)ELi(\d+)E", name) or re.search(r"conv_x3_kernel<(\d), (\d+)>", name)
    if m:
        return "conv_x3_kernel<%s>" % m.group(1) if m.group(2) == "0" else "conv_x3_kernel<%s,%s>" % m.groups()
    for k in ("conv_wgrad_hr_kernel", "conv_wgrad_thin_kernel", "conv_wgrad_kernel", "conv_thin_cout_kernel", "conv_thin_cin2_kernel", "conv_thin_cin_kernel",
              "conv_thin_tpd_kernel", "conv_thin_tp_kernel", "conv_thin_sc_kernel", "thin_tp_bwd_kernel", "epilogue_bwd_kernel", "unpack_wgrad_kernel",
              "head1_fwd_kernel", "head1_bwd_input_kernel", "bn_bwd_apply_kernel", "bn_bwd_reduce_kernel", "bn_apply_kernel", "channel_mean_sub_kernel", "round_weights_kernel"):
        if k in name:
            return k
    return name[:60]
