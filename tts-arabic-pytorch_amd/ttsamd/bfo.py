"""Kernel-level access to the bf16 octet engine (csrc/bfo*.hip, include/ttsamd.h ttsamd_bfo_*): layout converters,
weight packing and single layers.  Used by the parity tests and tools/bfo_bench.py; the model forwards reach the
same kernels through ttsamd_hifigan_forward under set_precision('bf16')."""
import ctypes as C

import numpy as np
import torch

from . import lib as L


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack(x, slope=1.0):
    """fp32 [B, C, L] (device) -> octet bf16 tensor, stored as int16 [B, C/8, L, 8], activated with leaky_relu(slope)."""
    x = x.contiguous().float()
    B, Cn, Ln = x.shape
    out = torch.empty(B, Cn // 8, Ln, 8, dtype=torch.int16, device=x.device)
    L.check(L.load().ttsamd_bfo_pack(_ptr(x), B, Cn, Ln, float(slope), _ptr(out), _stream()), 'bfo_pack')
    return out


def unpack(t, slope=1.0):
    """octet bf16 [B, C/8, L, 8] -> fp32 [B, C, L]; slope != 1 undoes the activation the tensor was stored with."""
    B, no, Ln, _ = t.shape
    out = torch.empty(B, no * 8, Ln, dtype=torch.float32, device=t.device)
    L.check(L.load().ttsamd_bfo_unpack(_ptr(t), B, no * 8, Ln, float(slope), _ptr(out), _stream()), 'bfo_unpack')
    return out


def pack_weight(w, up=1, device='cuda'):
    """torch Conv1d weight [Cout, Cin, K] (up = 1) or ConvTranspose1d weight [Cin, Cout, 2*up] -> packed bf16 on `device`."""
    w = np.ascontiguousarray(w.detach().cpu().float().numpy() if hasattr(w, 'detach') else w, dtype=np.float32)
    if up > 1:
        cin, cout, k = w.shape
    else:
        cout, cin, k = w.shape
    lib = L.load()
    n = lib.ttsamd_bfo_weight_elems(cout, cin, k, up)
    out = np.empty(n, dtype=np.uint16)
    L.check(lib.ttsamd_bfo_pack_weight(w.ctypes.data_as(C.c_void_p), cout, cin, k, up, out.ctypes.data_as(C.c_void_p)),
            'bfo_pack_weight')
    return torch.from_numpy(out.view(np.int16)).to(device)


def conv1d(x, wp, bias, cout, k, dilation=1, up=1, lens=None, len_mul=1, res=None, res_slope=1.0, sum_in=None, mode=0,
           div=1.0, out_slope=1.0, y=None, f32_out=False, res_f32=None):
    """f32_out: the result is an fp32 channel-first [B, cout, L] tensor (+ the fp32 channel-first residual res_f32)."""
    B, no, Ln, _ = x.shape
    if f32_out:
        yf = torch.zeros(B, cout, Ln, dtype=torch.float32, device=x.device) if y is None else y
        L.check(L.load().ttsamd_bfo_conv1d(_ptr(x), _ptr(wp), _ptr(bias), None, None, _ptr(lens), len_mul, B, no * 8, cout, k,
                                           dilation, up, Ln, 0, 1.0, 1.0, float(out_slope), None, _ptr(yf), _ptr(res_f32),
                                           _stream()), 'bfo_conv1d')
        return yf
    if y is None:
        y = torch.zeros(B, cout // 8, Ln * up, 8, dtype=torch.int16, device=x.device)
    L.check(L.load().ttsamd_bfo_conv1d(_ptr(x), _ptr(wp), _ptr(bias), _ptr(res), _ptr(sum_in), _ptr(lens), len_mul, B, no * 8,
                                       cout, k, dilation, up, Ln, mode, float(div), float(res_slope), float(out_slope),
                                       _ptr(y), None, None, _stream()), 'bfo_conv1d')
    return y


def resblock_pair(x, w1p, b1, w2p, b2, k, dilation, lens=None, len_mul=1, sum_in=None, mode=0, div=1.0, in_slope=0.1,
                  mid_slope=0.1, out_slope=0.1, y=None):
    B, no, Ln, _ = x.shape
    if y is None:
        y = torch.zeros_like(x)
    L.check(L.load().ttsamd_bfo_resblock_pair(_ptr(x), _ptr(w1p), _ptr(b1), _ptr(w2p), _ptr(b2), _ptr(sum_in), _ptr(lens),
                                              len_mul, B, no * 8, k, dilation, Ln, mode, float(div), float(in_slope),
                                              float(mid_slope), float(out_slope), _ptr(y), _stream()), 'bfo_resblock_pair')
    return y


def resblock_chain(x, w1p, b1, w2p, b2, dilations, lens=None, len_mul=1, sum_in=None, mode=0, div=1.0, in_slope=0.1, mid_slope=0.1,
                   out_slope=0.1, y=None, k=3):
    """A whole ResBlock (three pairs) in one launch, k = 3 or (C <= 64) k = 7; w1p / b1 / w2p / b2: lists of three device tensors."""
    import ctypes
    B, no, Ln, _ = x.shape
    if y is None:
        y = torch.zeros_like(x)
    arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    dl = (ctypes.c_int32 * 3)(*[int(d) for d in dilations])
    L.check(L.load().ttsamd_bfo_resblock_chain(_ptr(x), arr(w1p), arr(b1), arr(w2p), arr(b2), dl, _ptr(sum_in), _ptr(lens), len_mul,
                                               B, no * 8, Ln, mode, float(div), float(in_slope), float(mid_slope), float(out_slope),
                                               _ptr(y), _stream(), int(k)), 'bfo_resblock_chain')
    return y


def conv_post(x, w, bias, lens=None, len_mul=1):
    B, no, Ln, _ = x.shape
    wave = torch.zeros(B, Ln, dtype=torch.float32, device=x.device)
    L.check(L.load().ttsamd_bfo_conv_post(_ptr(x), _ptr(w), _ptr(bias), _ptr(lens), len_mul, B, no * 8, Ln, _ptr(wave), Ln,
                                          _stream()), 'bfo_conv_post')
    return wave


# ---- split-bf16 ("x3") mode: the same layers with every value = hi + lo (csrc/bfo3*.hip, ttsamd_bfo3_*) ----------------------
def pack3(x, slope=1.0):
    """fp32 [B, C, L] (device) -> x3 tensor, stored as int16 [B, C/8, L, 16] (32 bytes per octet and position: two halves of
    hi 4 | lo 4 bf16), activated with leaky_relu(slope)."""
    x = x.contiguous().float()
    B, Cn, Ln = x.shape
    out = torch.empty(B, Cn // 8, Ln, 16, dtype=torch.int16, device=x.device)
    L.check(L.load().ttsamd_bfo3_pack(_ptr(x), B, Cn, Ln, float(slope), _ptr(out), _stream()), 'bfo3_pack')
    return out


def unpack3(t, slope=1.0):
    """x3 tensor [B, C/8, L, 16] -> fp32 [B, C, L] (hi + lo); slope != 1 undoes the activation the tensor was stored with."""
    B, no, Ln, _ = t.shape
    out = torch.empty(B, no * 8, Ln, dtype=torch.float32, device=t.device)
    L.check(L.load().ttsamd_bfo3_unpack(_ptr(t), B, no * 8, Ln, float(slope), _ptr(out), _stream()), 'bfo3_unpack')
    return out


def pack_weight3(w, up=1, device='cuda'):
    """torch Conv1d weight [Cout, Cin, K] (up = 1) or ConvTranspose1d weight [Cin, Cout, 2*up] -> x3 weights on `device`."""
    w = np.ascontiguousarray(w.detach().cpu().float().numpy() if hasattr(w, 'detach') else w, dtype=np.float32)
    if up > 1:
        cin, cout, k = w.shape
    else:
        cout, cin, k = w.shape
    lib = L.load()
    n = lib.ttsamd_bfo3_weight_elems(cout, cin, k, up)
    out = np.empty(n, dtype=np.uint16)
    L.check(lib.ttsamd_bfo3_pack_weight(w.ctypes.data_as(C.c_void_p), cout, cin, k, up, out.ctypes.data_as(C.c_void_p)),
            'bfo3_pack_weight')
    return torch.from_numpy(out.view(np.int16)).to(device)


def conv1d3(x, wp, bias, cout, k, dilation=1, up=1, lens=None, len_mul=1, res=None, res_slope=1.0, sum_in=None, mode=0,
            div=1.0, out_slope=1.0, y=None, f32_out=False, res_f32=None):
    B, no, Ln, _ = x.shape
    if f32_out:
        yf = torch.zeros(B, cout, Ln, dtype=torch.float32, device=x.device) if y is None else y
        L.check(L.load().ttsamd_bfo3_conv1d(_ptr(x), _ptr(wp), _ptr(bias), None, None, _ptr(lens), len_mul, B, no * 8, cout, k,
                                            dilation, up, Ln, 0, 1.0, 1.0, float(out_slope), None, _ptr(yf), _ptr(res_f32),
                                            _stream()), 'bfo3_conv1d')
        return yf
    if y is None:
        y = torch.zeros(B, cout // 8, Ln * up, 16, dtype=torch.int16, device=x.device)
    L.check(L.load().ttsamd_bfo3_conv1d(_ptr(x), _ptr(wp), _ptr(bias), _ptr(res), _ptr(sum_in), _ptr(lens), len_mul, B, no * 8,
                                        cout, k, dilation, up, Ln, mode, float(div), float(res_slope), float(out_slope),
                                        _ptr(y), None, None, _stream()), 'bfo3_conv1d')
    return y


def resblock_pair3(x, w1p, b1, w2p, b2, k, dilation, lens=None, len_mul=1, sum_in=None, mode=0, div=1.0, in_slope=0.1,
                   mid_slope=0.1, out_slope=0.1, y=None):
    B, no, Ln, _ = x.shape
    if y is None:
        y = torch.zeros_like(x)
    L.check(L.load().ttsamd_bfo3_resblock_pair(_ptr(x), _ptr(w1p), _ptr(b1), _ptr(w2p), _ptr(b2), _ptr(sum_in), _ptr(lens),
                                               len_mul, B, no * 8, k, dilation, Ln, mode, float(div), float(in_slope),
                                               float(mid_slope), float(out_slope), _ptr(y), _stream()), 'bfo3_resblock_pair')
    return y


def resblock_chain3(x, w1p, b1, w2p, b2, dilations, lens=None, len_mul=1, sum_in=None, mode=0, div=1.0, in_slope=0.1, mid_slope=0.1,
                    out_slope=0.1, y=None):
    """A whole k = 3 ResBlock (three pairs) in one launch of the split-bf16 engine; w1p / b1 / w2p / b2: lists of three device tensors."""
    import ctypes
    B, no, Ln, _ = x.shape
    if y is None:
        y = torch.zeros_like(x)
    arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    dl = (ctypes.c_int32 * 3)(*[int(d) for d in dilations])
    L.check(L.load().ttsamd_bfo3_resblock_chain(_ptr(x), arr(w1p), arr(b1), arr(w2p), arr(b2), dl, _ptr(sum_in), _ptr(lens), len_mul,
                                                B, no * 8, Ln, mode, float(div), float(in_slope), float(mid_slope), float(out_slope),
                                                _ptr(y), _stream()), 'bfo3_resblock_chain')
    return y


def conv_post3(x, w, bias, lens=None, len_mul=1):
    B, no, Ln, _ = x.shape
    wave = torch.zeros(B, Ln, dtype=torch.float32, device=x.device)
    L.check(L.load().ttsamd_bfo3_conv_post(_ptr(x), _ptr(w), _ptr(bias), _ptr(lens), len_mul, B, no * 8, Ln, _ptr(wave), Ln,
                                           _stream()), 'bfo3_conv_post')
    return wave
