"""Tensor-level wrappers over the C ABI (include/gsseg.h).  Shapes/dtypes are validated HERE, before
any launch, so failures are synchronous Python exceptions.  torch supplies device memory and the
current HIP stream only; all arithmetic happens in libgsseg_hip.so."""
from __future__ import annotations

import collections
import ctypes
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import ACT_NONE, GS_BF16, GS_F16, GsConvGeom

DTYPES = {"f16": (GS_F16, torch.float16), "bf16": (GS_BF16, torch.bfloat16)}


def dt_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float16:
        return GS_F16
    if t.dtype == torch.bfloat16:
        return GS_BF16
    raise TypeError(f"expected a float16/bfloat16 tensor, got {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_RAW_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Handle of torch's current HIP stream on the current device.  torch.cuda.current_stream() builds a Stream object
    (~9 us per call, a third of the host time of a small-batch step); the raw accessors return the same handle."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return _RAW_STREAM(_RAW_DEVICE())
    return torch.cuda.current_stream().cuda_stream


_TAP_ARRAYS = {}


def _tap_arrays(taps):
    """(dy, dx) int32[9] ctypes arrays of a 3x3 tap list, built once per list (2.2 us per call otherwise: ~50 conv launches per
    U-Net pass at the script's batch 2, where the host bounds the step)."""
    key = tuple(taps)
    r = _TAP_ARRAYS.get(key)
    if r is None:
        r = _TAP_ARRAYS[key] = ((ctypes.c_int32 * 9)(*[t[0] for t in key]), (ctypes.c_int32 * 9)(*[t[1] for t in key]))
    return r


_TAPS3D = {sgn: ((ctypes.c_int32 * 3)(*[sgn * (k - 1) for k in range(3)]), (ctypes.c_int32 * 9)(*[sgn * (k // 3 - 1) for k in range(9)]),
                 (ctypes.c_int32 * 9)(*[sgn * (k % 3 - 1) for k in range(9)])) for sgn in (1, -1)}


def _dev(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("semantic_segmentation_amd ops need tensors on the MI355X (cuda) device; "
                           "there is no CPU path")
    # launches go to the CURRENT device's current stream: a tensor of another device would be dereferenced there
    cur = _RAW_DEVICE() if _RAW_DEVICE is not None else torch.cuda.current_device()
    if t.device.index != cur:
        raise RuntimeError(f"tensor on cuda:{t.device.index} but the current device is cuda:{cur}: one process per GPU "
                           "-- call torch.cuda.set_device (or run under torch.cuda.device(t.device))")


def _f32(t: Optional[torch.Tensor], name: str):
    if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
        raise TypeError(f"{name} must be a contiguous float32 tensor")


# ---------------------------------------------------------------------------- geometry builders
def make_geom(N, IH, IW, Cin, OHg, OWg, Cout, OH, OW, taps: Sequence[Tuple[int, int]], isy=1, isx=1,
              osy=1, osx=1, ooy=0, oox=0, in_stride=None, in_coff=0, out_stride=None, out_coff=0,
              tap_w: Optional[Sequence[int]] = None, Dg=1, Din=1, Dout=1, isz=1, osz=1, ooz=0,
              tap_dz: Optional[Sequence[int]] = None) -> GsConvGeom:
    g = GsConvGeom()
    g.N, g.IH, g.IW, g.Cin = N, IH, IW, Cin
    g.in_pix_stride = Cin if in_stride is None else in_stride
    g.in_coff = in_coff
    g.OHg, g.OWg, g.Cout, g.OH, g.OW = OHg, OWg, Cout, OH, OW
    g.out_pix_stride = Cout if out_stride is None else out_stride
    g.out_coff = out_coff
    g.isy, g.isx, g.osy, g.osx, g.ooy, g.oox = isy, isx, osy, osx, ooy, oox
    if not 0 < len(taps) <= _lib.GS_MAX_TAPS:
        raise ValueError(f"{len(taps)} taps out of range")
    g.ntaps = len(taps)
    for i, (dy, dx) in enumerate(taps):
        g.tap_dy[i], g.tap_dx[i] = dy, dx
        g.tap_w[i] = i if tap_w is None else tap_w[i]
        g.tap_dz[i] = 0 if tap_dz is None else tap_dz[i]
    g.Dg, g.Din, g.Dout, g.isz, g.osz, g.ooz = Dg, Din, Dout, isz, osz, ooz
    return g


def conv_out_size(i: int, k: int, s: int, p: int) -> int:
    return (i + 2 * p - k) // s + 1


def geom_conv(N, IH, IW, Cin, Cout, k, stride, pad, **kw) -> GsConvGeom:
    """nn.Conv2d(k, stride, pad): taps in (ky,kx) row-major order, weights [k*k][Cout][Cin]."""
    OH, OW = conv_out_size(IH, k, stride, pad), conv_out_size(IW, k, stride, pad)
    taps = [(ky - pad, kx - pad) for ky in range(k) for kx in range(k)]
    return make_geom(N, IH, IW, Cin, OH, OW, Cout, OH, OW, taps, isy=stride, isx=stride, **kw)


def geom_conv_dgrad_s1(N, IH, IW, Cin, Cout, k, pad, **kw) -> GsConvGeom:
    """Data gradient of a stride-1 Conv2d as a convolution over dy: dx[i] = sum_t dy[i + pad - k_t] Wd[t].
    'Input' of this launch is dy [N,OH,OW,Cout]; 'output' is dx [N,IH,IW,Cin]; weights [k*k][Cin][Cout]."""
    OH, OW = conv_out_size(IH, k, 1, pad), conv_out_size(IW, k, 1, pad)
    taps = [(pad - ky, pad - kx) for ky in range(k) for kx in range(k)]
    return make_geom(N, OH, OW, Cout, IH, IW, Cin, IH, IW, taps, **kw)


# ---------------------------------------------------------------------------- per-launch timing
class KernelTimer:
    """Brackets selected launches with HIP events on the launch stream (torch's current stream, which is
    the stream handed to the C ABI) and accumulates algorithmic work per kernel class.  Used by bench.py
    for the roofline figures; off by default (TIMER is None)."""

    def __init__(self, pool: int = 0):
        self.records = []          # (kind, flops, bytes, ev0, ev1)
        # events created (and recorded once, which is what really creates the HIP event) up front, so that the timed
        # region of a benchmark does not contain thousands of hipEventCreate calls
        self._pool = [torch.cuda.Event(enable_timing=True) for _ in range(pool)]
        for ev in self._pool:
            ev.record()
        if pool:
            torch.cuda.synchronize()

    def _event(self):
        return self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)

    def start(self):
        ev = self._event()
        ev.record()
        return ev

    def stop(self, kind, ev0, flops=0.0, nbytes=0.0):
        ev1 = self._event()
        ev1.record()
        self.records.append((kind, float(flops), float(nbytes), ev0, ev1))

    def reset(self):
        """forget the launches bracketed so far (warm-up steps run with the timer attached); the pool keeps its unused events"""
        self.records = []

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, fl, nb, e0, e1 in self.records:
            d = out.setdefault(kind, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
        return out


TIMER: Optional[KernelTimer] = None


def _geom_flops(g: GsConvGeom) -> float:
    return 2.0 * g.N * g.Dg * g.OHg * g.OWg * g.Cout * g.ntaps * g.Cin


def _geom_bytes(g: GsConvGeom, wgrad: bool = False) -> float:
    """algorithmic HBM bytes of one implicit-GEMM launch: the input tensor once, the output once, the weights once (fp32 for a
    weight gradient)"""
    inp = 2.0 * g.N * g.Din * g.IH * g.IW * g.Cin
    out = 2.0 * g.N * g.Dg * g.OHg * g.OWg * g.Cout
    return inp + out + (4.0 if wgrad else 2.0) * g.ntaps * g.Cin * g.Cout


def geom_convT_class(N, IH, IW, Cin, Cout, k, pad, py, px, OH=None, OW=None, **kw) -> GsConvGeom:
    """Sub-pixel class (py,px) of ConvTranspose2d(k, stride 2, pad): output pixels (2i+py, 2j+px) <- input
    (i+dy, j+dx) for the taps with ky == py+pad (mod 2); weight slots index a [k*k][Cout][Cin] pack."""
    kys = [ky for ky in range(k) if (py + pad - ky) % 2 == 0]
    kxs = [kx for kx in range(k) if (px + pad - kx) % 2 == 0]
    taps = [((py + pad - ky) // 2, (px + pad - kx) // 2) for ky in kys for kx in kxs]
    slots = [ky * k + kx for ky in kys for kx in kxs]
    OHf = (IH - 1) * 2 - 2 * pad + k if OH is None else OH
    OWf = (IW - 1) * 2 - 2 * pad + k if OW is None else OW
    ohg, owg = (OHf - py + 1) // 2, (OWf - px + 1) // 2
    return make_geom(N, IH, IW, Cin, ohg, owg, Cout, OHf, OWf, taps, osy=2, osx=2, ooy=py, oox=px, tap_w=slots, **kw)


def geom_conv_s2_dgrad_class(N, IH, IW, Cin, Cout, k, pad, py, px, **kw) -> GsConvGeom:
    """Data gradient of Conv2d(k, stride 2, pad) for input pixels (2i+py, 2j+px): reads dy [N,OH,OW,Cout] at
    (i + (py+pad-ky)/2, ...) for ky == py+pad (mod 2); weight slots index the [k*k][Cin][Cout] dgrad pack."""
    OH, OW = conv_out_size(IH, k, 2, pad), conv_out_size(IW, k, 2, pad)
    kys = [ky for ky in range(k) if (py + pad - ky) % 2 == 0]
    kxs = [kx for kx in range(k) if (px + pad - kx) % 2 == 0]
    taps = [((py + pad - ky) // 2, (px + pad - kx) // 2) for ky in kys for kx in kxs]
    slots = [ky * k + kx for ky in kys for kx in kxs]
    ihg, iwg = (IH - py + 1) // 2, (IW - px + 1) // 2
    return make_geom(N, OH, OW, Cout, ihg, iwg, Cin, IH, IW, taps, osy=2, osx=2, ooy=py, oox=px, tap_w=slots, **kw)


# ---------------------------------------------------------------------------- MFMA engine
_SPLITK_WS = collections.OrderedDict()      # (device index, stream handle) -> zeroed fp32 workspace, least recently used first
_SPLITK_WS_MAX = int(os.environ.get("GSSEG_SPLITK_WS_MAX", "3"))
_SPLITK_PINNED = set()                      # keys baked into a captured graph: never evicted


def _splitk_workspace(device: torch.device, stream) -> Optional[torch.Tensor]:
    """gs_conv_igemm's split-K workspace (134 MB), passed with every call: one per (device, stream), because launches that
    share one must be ordered on one stream (ticket counters and slabs are reused); allocated zeroed once, the kernels
    leave the counters zeroed.  At most GSSEG_SPLITK_WS_MAX (3) un-pinned ones are kept: a process that runs the engine on many
    streams (or whose stream handles torch recycles) evicts the least recently used one -- it was allocated and used on its own
    stream only, so handing it back to torch's stream-ordered allocator is safe -- instead of pinning 134 MB per handle.
    Stream capture: a workspace handed to a capturing stream is baked into the graph, so it is PINNED (never evicted) from
    then on, and it must exist BEFORE the capture starts (allocating it here would put the buffer into the graph's pool and its
    zero fill into the graph, while this cache would go on handing the pointer to eager code): graphs.capture_step and
    harness.EndToEndTrainer create it first; anything else that captures engine code calls ops.splitk_workspace_for_capture()."""
    if os.environ.get("GSSEG_SPLITK", "1") == "0":
        return None
    key = (device.index, int(stream or 0))
    capturing = torch.cuda.is_current_stream_capturing()
    ws = _SPLITK_WS.get(key)
    if ws is None:
        if capturing:
            raise RuntimeError("the split-K workspace of this stream does not exist yet and cannot be created during a stream capture: "
                               "call semantic_segmentation_amd.ops.splitk_workspace_for_capture() on the capture stream before capturing")
        n = int(_lib.load().gs_conv_igemm_workspace_floats())
        ws = torch.zeros(n, dtype=torch.float32, device=device)     # zero fill runs on this same (current) stream
        _SPLITK_WS[key] = ws
        evictable = [k for k in _SPLITK_WS if k not in _SPLITK_PINNED and k != key]
        while len(evictable) + 1 > max(1, _SPLITK_WS_MAX) and evictable:
            _SPLITK_WS.pop(evictable.pop(0))
    else:
        _SPLITK_WS.move_to_end(key)
    if capturing:
        _SPLITK_PINNED.add(key)
    return ws


def splitk_workspace_for_capture(device: Optional[torch.device] = None) -> Optional[torch.Tensor]:
    """Create (eagerly) and pin the split-K workspace of the CURRENT stream: call on the capture stream before a stream capture of
    engine code (a hipGraph bakes the pointer in; see _splitk_workspace)."""
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    stream = _stream()
    ws = _splitk_workspace(device, stream)
    if ws is not None:
        _SPLITK_PINNED.add((device.index, int(stream or 0)))
    return ws


def conv_igemm(g: GsConvGeom, x, w, y, bias=None, bn_partials=None, act=ACT_NONE):
    _dev(x)
    _f32(bias, "bias"); _f32(bn_partials, "bn_partials")
    if not (x.dtype == w.dtype == y.dtype):
        raise TypeError("conv_igemm: x, w, y must share one 16-bit dtype")
    if bn_partials is not None:
        need = _lib.load().gs_bn_partials_floats(conv_igemm_mtiles(g), g.Cout)
        if bn_partials.numel() < need:
            raise ValueError(f"bn_partials too small: {bn_partials.numel()} < {need}")
    stream = _stream()
    ws = _splitk_workspace(x.device, stream)
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv_igemm", g, _p(x), _p(w), _p(y), _p(bias), _p(bn_partials), act, dt_code(x), _p(ws),
              ws.numel() if ws is not None else 0, stream)
    if ev is not None:
        TIMER.stop("igemm_fwd", ev, _geom_flops(g), _geom_bytes(g))


# taps of a 3x3/pad-1 conv in weight order, and the flipped list that turns the same kernel into its dgrad
def conv_igemm_batch(geoms, x, ws_list, y, bias=None, bn_partials=None, act=ACT_NONE):
    """Up to four GEMMs (the sub-pixel classes of one layer) over the same x / y / bias in ONE launch: geoms[i] with weights
    ws_list[i] and optional BatchNorm partials bn_partials[i].  At the script's batch size (2) a class launch fills a
    fraction of the chip; together the classes fill it and pay one launch gap."""
    _dev(x)
    _f32(bias, "bias")
    n = len(geoms)
    if not 1 <= n <= 4 or len(ws_list) != n or (bn_partials is not None and len(bn_partials) != n):
        raise ValueError("conv_igemm_batch: 1..4 GEMMs with one weight tensor (and optionally one partials tensor) each")
    if any(w.dtype != x.dtype for w in ws_list) or y.dtype != x.dtype:
        raise TypeError("conv_igemm_batch: x, w, y must share one 16-bit dtype")
    if bn_partials is not None:
        for g, p in zip(geoms, bn_partials):
            _f32(p, "bn_partials")
            if p.numel() < _lib.load().gs_bn_partials_floats(conv_igemm_mtiles(g), g.Cout):
                raise ValueError("conv_igemm_batch: bn_partials too small")
    garr = (ctypes.POINTER(GsConvGeom) * n)(*[ctypes.pointer(g) for g in geoms])
    warr = (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws_list])
    parr = (ctypes.c_void_p * n)(*[p.data_ptr() for p in bn_partials]) if bn_partials is not None else None
    stream = _stream()
    ws = _splitk_workspace(x.device, stream)
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv_igemm_batch", n, garr, _p(x), warr, _p(y), _p(bias), parr, act, dt_code(x), _p(ws),
              ws.numel() if ws is not None else 0, stream)
    if ev is not None:
        TIMER.stop("igemm_fwd", ev, sum(_geom_flops(g) for g in geoms), sum(_geom_bytes(g) for g in geoms))


TAPS3_FWD = [(ky - 1, kx - 1) for ky in range(3) for kx in range(3)]
TAPS3_DGRAD = [(1 - ky, 1 - kx) for ky in range(3) for kx in range(3)]
USE_HALO_CONV = os.environ.get("GSSEG_CONV3X3", "halo") != "generic"


def conv3x3_mtiles(N, H, W, Cout) -> int:
    return _lib.load().gs_conv3x3_mtiles(N, H, W, Cout)


def conv3x3_stat_rows(N, H, W, Cin, Cout, pair=False) -> int:
    """Rows of BatchNorm partial sums a conv3x3 launch of these dimensions WRITES (what bn_finalize / bn_partials_colsum must
    be told): one per block and cout-tile group on the LDS-DMA kernel, one per patch otherwise and in the pair forward.  A
    buffer of conv3x3_mtiles() rows always suffices."""
    # pair: False / True (the pair forward: Cin = its K extent) / "q" (a "q" stage: Cin = K = 2 * channels; its form is chosen per shape)
    return int(_lib.load().gs_conv3x3_stat_rows(N, H, W, Cin, Cout, 2 if pair == "q" else (1 if pair else 0)))


def conv3d3_stat_rows(NB, D, H, W, Cin, Cout) -> int:
    return conv3x3_stat_rows(NB * D, H, W, Cin, Cout)


def set_persistent_grid(blocks: int = 0) -> None:
    """At most `blocks` workgroups for the persistent conv kernels (0: the default, one per CU); parallel.py leaves CUs to RCCL."""
    _lib.call("gs_set_persistent_grid", int(blocks))


def get_persistent_grid() -> int:
    return int(_lib.load().gs_get_persistent_grid())


def conv3x3_set_kernel_form(form: int = -1) -> None:
    """Diagnostics: pin the form of the conv3x3 kernel (-1 auto, 0 register-staged, 4 / 8 LDS-DMA waves per block).
    Process-wide; tests compare the forms on the same operands and restore -1."""
    _lib.call("gs_conv3x3_set_kernel_form", int(form))


def conv3x3(x, w, y, N, H, W, Cin, Cout, taps=TAPS3_FWD, bias=None, bn_partials=None, act=ACT_NONE,
            in_stride=None, in_coff=0, out_stride=None, out_coff=0):
    """3x3/s1/p1 convolution (or its data gradient with TAPS3_DGRAD + the dgrad weight pack) on the
    halo-reuse MFMA kernel.  x/y NHWC 16-bit (strided), w [9][Cout][Cin]."""
    _dev(x)
    _f32(bias, "bias"); _f32(bn_partials, "bn_partials")
    if not (x.dtype == w.dtype == y.dtype):
        raise TypeError("conv3x3: x, w, y must share one 16-bit dtype")
    if bn_partials is not None and bn_partials.numel() < bn_partials_numel(conv3x3_stat_rows(N, H, W, Cin, Cout), Cout):
        raise ValueError("conv3x3: bn_partials too small (conv3x3_stat_rows rows of [2][Cout])")
    dy, dx = _tap_arrays(taps)
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv3x3", _p(x), _p(w), _p(y), _p(bias), _p(bn_partials), N, H, W, Cin,
              Cin if in_stride is None else in_stride, in_coff, Cout, Cout if out_stride is None else out_stride,
              out_coff, dy, dx, act, dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("conv3x3_halo", ev, 2.0 * N * H * W * Cout * 9 * Cin,
                   2.0 * (N * H * W * (Cin + Cout) + 9 * Cin * Cout))


def conv3x3_wgrad(x, dy, dw, N, H, W, Cin, Cout, in_stride=None, in_coff=0, out_stride=None, out_coff=0):
    """dw[9][Cout][Cin] (fp32, caller zeroes) += weight gradient of the 3x3/s1/p1 conv, halo-reuse MFMA kernel."""
    _dev(x)
    _f32(dw, "dw")
    if x.dtype != dy.dtype:
        raise TypeError("conv3x3_wgrad: x and dy must share one 16-bit dtype")
    if dw.numel() < 9 * Cout * Cin:
        raise ValueError("conv3x3_wgrad: dw too small")
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv3x3_wgrad", _p(x), _p(dy), _p(dw), N, H, W, Cin, Cin if in_stride is None else in_stride,
              in_coff, Cout, Cout if out_stride is None else out_stride, out_coff, dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("wgrad3x3_halo", ev, 2.0 * N * H * W * Cout * 9 * Cin,
                   2.0 * N * H * W * (Cin + Cout) + 4.0 * 9 * Cin * Cout)


def upconv2x2_fwd(x, w, bias, y, N, D, IH, IW, Cin, Cout, Dout, OH, OW, in_stride=None, in_coff=0, out_stride=None,
                  out_coff=0, ooz=0, ooy=0, oox=0, act=ACT_NONE):
    """ConvTranspose2d/3d(kernel 2, stride 2) forward as one pointwise MFMA GEMM with a sub-pixel scatter.
    w: packed [4|8][Cout][Cin] (pack_weight(..., transposed=True)); y may be a concat buffer (out_coff)."""
    _dev(x)
    _f32(bias, "bias")
    if not (x.dtype == w.dtype == y.dtype):
        raise TypeError("upconv2x2_fwd: x, w, y must share one 16-bit dtype")
    ncls = 8 if (D > 1 or Dout > 1) else 4
    if w.numel() != ncls * Cout * Cin:
        raise ValueError("upconv2x2_fwd: packed weight has the wrong size")
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_upconv2x2_fwd", _p(x), _p(w), _p(bias), _p(y), N, D, IH, IW, Cin,
              Cin if in_stride is None else in_stride, in_coff, Cout, Dout, OH, OW,
              Cout if out_stride is None else out_stride, out_coff, ooz, ooy, oox, act, dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("igemm_fwd", ev, 2.0 * N * D * IH * IW * Cin * ncls * Cout,
                   2.0 * (N * D * IH * IW * (Cin + ncls * Cout) + ncls * Cin * Cout))


def upconv2x2_dgrad(geom, dy, wd, dx, N, IH, IW, Cin, Cout, OH, OW, dy_stride, dy_coff, ooy, oox):
    """Data gradient of ConvTranspose2d(k 2, s 2): the LDS-DMA GEMM over (sub-pixel class, co) where it covers the shape
    (gs_upconv2x2_dgrad), else the generic engine on `geom` (the 4-tap stride-2 geometry of the same layer).  Both are HIP
    kernels; dx dense [N, IH, IW, Cin]."""
    _dev(dy)
    if not (dy.dtype == wd.dtype == dx.dtype):
        raise TypeError("upconv2x2_dgrad: dy, wd, dx must share one 16-bit dtype")
    ev = TIMER.start() if TIMER is not None else None
    rc = _lib.load().gs_upconv2x2_dgrad(_p(dy), _p(wd), _p(dx), N, IH, IW, Cin, Cout, OH, OW, dy_stride, dy_coff, ooy, oox,
                                        Cin, 0, dt_code(dy), _stream())
    if rc == _lib.GS_EUNSUPPORTED:            # shape outside the DMA GEMM: the generic engine (it times itself)
        return conv_igemm(geom, dy, wd, dx)
    if rc != 0:
        _lib.check(rc, "gs_upconv2x2_dgrad")
    if ev is not None:
        TIMER.stop("igemm_fwd", ev, 2.0 * N * IH * IW * Cin * 4 * Cout,
                   2.0 * (N * IH * IW * (Cin + 4 * Cout) + 4 * Cin * Cout))


def conv3d3_eligible(Cin, Cout, out_stride=None, out_coff=0) -> bool:
    """shapes the 3-D halo kernels take (others go through conv_igemm with depth taps)"""
    os_ = Cout if out_stride is None else out_stride
    return USE_HALO_CONV and Cin % 8 == 0 and Cout % 8 == 0 and os_ % 8 == 0 and out_coff % 8 == 0


def conv3d3_mtiles(NB, D, H, W, Cout) -> int:
    return int(_lib.load().gs_conv3d_3x3x3_mtiles(NB, D, H, W, Cout))


def conv3d3(x, w, y, NB, D, H, W, Cin, Cout, dgrad=False, bias=None, bn_partials=None, act=ACT_NONE, in_stride=None,
            in_coff=0, out_stride=None, out_coff=0):
    """Conv3d(k3, p1) forward (w = [27][Cout][Cin] pack) or data gradient (dgrad=True: x = dY, w = [27][Cin][Cout]
    pack, Cin/Cout given from the kernel's point of view: Cin = channels of x) on depth-slice NHWC tensors."""
    _dev(x)
    _f32(bias, "bias"); _f32(bn_partials, "bn_partials")
    if not (x.dtype == w.dtype == y.dtype):
        raise TypeError("conv3d3: x, w, y must share one 16-bit dtype")
    if w.numel() != 27 * Cout * Cin:
        raise ValueError("conv3d3: packed weight has the wrong size")
    if bn_partials is not None and bn_partials.numel() < bn_partials_numel(conv3d3_stat_rows(NB, D, H, W, Cin, Cout), Cout):
        raise ValueError("conv3d3: bn_partials too small")
    sgn = -1 if dgrad else 1
    dz, dy, dx = _TAPS3D[sgn]
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv3d_3x3x3", _p(x), _p(w), _p(y), _p(bias), _p(bn_partials), NB, D, H, W, Cin,
              Cin if in_stride is None else in_stride, in_coff, Cout, Cout if out_stride is None else out_stride, out_coff,
              dz, dy, dx, act, dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("conv3x3_halo", ev, 2.0 * NB * D * H * W * Cout * 27 * Cin)


def conv3d3_wgrad(x, dy, dw, NB, D, H, W, Cin, Cout, in_stride=None, in_coff=0, out_stride=None, out_coff=0):
    """dw[27][Cout][Cin] (fp32, caller zeroes) += weight gradient of Conv3d(k3, p1)."""
    _dev(x)
    _f32(dw, "dw")
    if x.dtype != dy.dtype:
        raise TypeError("conv3d3_wgrad: x and dy must share one 16-bit dtype")
    if dw.numel() < 27 * Cout * Cin:
        raise ValueError("conv3d3_wgrad: dw too small")
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv3d_3x3x3_wgrad", _p(x), _p(dy), _p(dw), NB, D, H, W, Cin, Cin if in_stride is None else in_stride,
              in_coff, Cout, Cout if out_stride is None else out_stride, out_coff, dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("wgrad3x3_halo", ev, 2.0 * NB * D * H * W * Cout * 27 * Cin)


def conv3d3_wgrad_ws_floats(NB, D, H, W, Cin, Cout) -> int:
    return int(_lib.load().gs_conv3d_3x3x3_wgrad_ws_floats(NB, D, H, W, Cin, Cout))


def conv3d3_wgrad_det(x, dy, ws, grad, NB, D, H, W, Cin, Cout, gscale, in_stride=None, in_coff=0, out_stride=None,
                      out_coff=0):
    """Deterministic Conv3d(k3,p1) weight gradient into the reference layout [Cout][Cin][3][3][3] (27 taps fastest):
    split-K parts in fp32 slabs (no zero fill), ordered reduction fused with scale + unpack.  No atomics."""
    _dev(x)
    _f32(ws, "ws"); _f32(grad, "grad")
    if x.dtype != dy.dtype:
        raise TypeError("conv3d3_wgrad_det: x and dy must share one 16-bit dtype")
    if ws.numel() < conv3d3_wgrad_ws_floats(NB, D, H, W, Cin, Cout) or grad.numel() != 27 * Cout * Cin:
        raise ValueError("conv3d3_wgrad_det: workspace / gradient size")
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv3d_3x3x3_wgrad_slabs", _p(x), _p(dy), _p(ws), NB, D, H, W, Cin, Cin if in_stride is None else in_stride,
              in_coff, Cout, Cout if out_stride is None else out_stride, out_coff, dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("wgrad3x3_halo", ev, 2.0 * NB * D * H * W * Cout * 27 * Cin)
    parts = int(_lib.load().gs_conv3d_3x3x3_wgrad_parts(NB, D, H, W, Cin, Cout))
    _lib.call("gs_wgrad_reduce_unpack", _p(ws), parts, _p(grad), Cout, Cin, 27, 0, float(gscale), _stream())


def upsample2x_bilinear_fwd(x, y, N, IH, IW, C, OH, OW, in_stride=None, in_coff=0, out_stride=None, out_coff=0,
                            ooy=0, oox=0):
    """nn.Upsample(scale_factor=2, bilinear, align_corners=True) into a channel slice of the [N,OH,OW,*] buffer y."""
    _dev(x)
    if x.dtype != y.dtype:
        raise TypeError("upsample2x_bilinear_fwd: x and y must share one 16-bit dtype")
    _lib.call("gs_upsample2x_bilinear_fwd", _p(x), _p(y), N, IH, IW, C, C if in_stride is None else in_stride, in_coff,
              OH, OW, C if out_stride is None else out_stride, out_coff, ooy, oox, dt_code(x), _stream())


def upsample2x_bilinear_bwd(dy, dx, N, IH, IW, C, OH, OW, dy_stride=None, dy_coff=0, dx_stride=None, dx_coff=0,
                            ooy=0, oox=0):
    _dev(dy)
    if dy.dtype != dx.dtype:
        raise TypeError("upsample2x_bilinear_bwd: dy and dx must share one 16-bit dtype")
    _lib.call("gs_upsample2x_bilinear_bwd", _p(dy), _p(dx), N, IH, IW, C, C if dy_stride is None else dy_stride, dy_coff,
              OH, OW, C if dx_stride is None else dx_stride, dx_coff, ooy, oox, dt_code(dy), _stream())


def conv3x3_wgrad_parts(N, H, W, Cin, Cout) -> int:
    return int(_lib.load().gs_conv3x3_wgrad_parts(N, H, W, Cin, Cout))


def conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout) -> int:
    return int(_lib.load().gs_conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout))


def conv3x3_wgrad_det(x, dy, ws, grad, N, H, W, Cin, Cout, gscale, in_stride=None, in_coff=0, out_stride=None,
                      out_coff=0):
    """Deterministic 3x3 weight gradient straight into the reference layout: split-K parts in fp32 slabs (ws, no zero
    fill), then an ordered reduction fused with the scale + [Cout][Cin][3][3] unpack.  No atomics."""
    _dev(x)
    _f32(ws, "ws"); _f32(grad, "grad")
    if x.dtype != dy.dtype:
        raise TypeError("conv3x3_wgrad_det: x and dy must share one 16-bit dtype")
    if ws.numel() < conv3x3_wgrad_ws_floats(N, H, W, Cin, Cout) or grad.numel() != 9 * Cout * Cin:
        raise ValueError("conv3x3_wgrad_det: workspace / gradient size")
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv3x3_wgrad_slabs", _p(x), _p(dy), _p(ws), N, H, W, Cin, Cin if in_stride is None else in_stride,
              in_coff, Cout, Cout if out_stride is None else out_stride, out_coff, dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("wgrad3x3_halo", ev, 2.0 * N * H * W * Cout * 9 * Cin,
                   2.0 * N * H * W * (Cin + Cout) + 4.0 * 9 * Cin * Cout)
    parts = int(_lib.load().gs_conv3x3_wgrad_parts(N, H, W, Cin, Cout))
    _lib.call("gs_wgrad_reduce_unpack", _p(ws), parts, _p(grad), Cout, Cin, 9, 0, float(gscale), _stream())


def conv_igemm_mtiles(g: GsConvGeom) -> int:
    return _lib.load().gs_conv_igemm_mtiles(g)


def conv_wgrad_single_pass(g: GsConvGeom) -> bool:
    """True when the weight gradient of this geometry is computed without a K split, i.e. conv_wgrad(assign=True) may
    write into an un-zeroed dw."""
    return bool(_lib.load().gs_conv_wgrad_single_pass(g))


def conv_wgrad(g: GsConvGeom, x, dy, dw, assign: bool = False):
    """dw[tap][Cout][Cin] (fp32) += weight gradient (caller zeroes dw); assign=True writes instead (single-pass only)."""
    _dev(x)
    _f32(dw, "dw")
    if x.dtype != dy.dtype:
        raise TypeError("conv_wgrad: x and dy must share one 16-bit dtype")
    if dw.numel() < g.ntaps * g.Cout * g.Cin:
        raise ValueError("conv_wgrad: dw too small")
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv_wgrad_assign" if assign else "gs_conv_wgrad", g, _p(x), _p(dy), _p(dw), dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("igemm_wgrad", ev, _geom_flops(g), _geom_bytes(g, True))


def conv_wgrad_parts(g: GsConvGeom) -> int:
    return int(_lib.load().gs_conv_wgrad_parts(g))


def conv_wgrad_ws_floats(g: GsConvGeom) -> int:
    return int(_lib.load().gs_conv_wgrad_ws_floats(g))


def conv_wgrad_det(g: GsConvGeom, x, dy, ws, grad, A, B, taps, gscale, transposed=False, packed=False):
    """Deterministic generic weight gradient straight into the reference layout grad[A][B][taps] (transposed: [B][A][taps];
    packed: the kernel layout [taps][A][B] is kept): K parts in fp32 slabs (ws, no zero fill, no atomics) + the ordered
    reduction fused with scale and unpack."""
    _dev(x)
    _f32(ws, "ws"); _f32(grad, "grad")
    if x.dtype != dy.dtype:
        raise TypeError("conv_wgrad_det: x and dy must share one 16-bit dtype")
    if A != g.Cout or B != g.Cin or grad.numel() != taps * A * B or ws.numel() < conv_wgrad_ws_floats(g):
        raise ValueError("conv_wgrad_det: workspace / gradient size")
    if packed:
        A, taps = taps * A, 1              # [taps*A][B][1] is the slab layout itself: a plain ordered sum
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv_wgrad_slabs", g, _p(x), _p(dy), _p(ws), dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("igemm_wgrad", ev, _geom_flops(g), _geom_bytes(g, True))
    parts = int(_lib.load().gs_conv_wgrad_parts(g))
    _lib.call("gs_wgrad_reduce_unpack", _p(ws), parts, _p(grad), A, B, taps, 1 if transposed else 0, float(gscale), _stream())


def upconv2x2_wgrad_ws_floats(N, IH, IW, Cin, Cout) -> int:
    """workspace floats of upconv2x2_wgrad_det's LDS-DMA path (0: shape not covered, the generic engine is used)"""
    return int(_lib.load().gs_upconv2x2_wgrad_ws_floats(N, IH, IW, Cin, Cout))


def upconv2x2_wgrad_det(geom, x, dy, ws, grad, N, IH, IW, Cin, Cout, OH, OW, x_stride, dy_stride, dy_coff, ooy, oox, gscale):
    """Deterministic weight gradient of ConvTranspose2d(k 2, s 2) into grad [Cin][Cout][2][2]: x = the layer's input (pixel
    stride x_stride), dy = d(output) inside a [N,OH,OW,dy_stride] buffer at channel dy_coff.  The LDS-DMA pointwise GEMM
    (csrc/upwgrad.hip) where it covers the shape, else the generic engine on `geom` (the layer seen from its output side)."""
    _dev(x)
    _f32(ws, "ws"); _f32(grad, "grad")
    if x.dtype != dy.dtype:
        raise TypeError("upconv2x2_wgrad_det: x and dy must share one 16-bit dtype")
    lib = _lib.load()
    parts = int(lib.gs_upconv2x2_wgrad_parts(N, IH, IW, Cin, Cout))
    if parts > 0 and grad.numel() == 4 * Cin * Cout and ws.numel() >= parts * 4 * Cin * Cout:
        ev = TIMER.start() if TIMER is not None else None
        rc = lib.gs_upconv2x2_wgrad_slabs(_p(x), _p(dy), _p(ws), N, IH, IW, Cin, x_stride, 0, Cout, OH, OW, dy_stride, dy_coff,
                                          ooy, oox, dt_code(x), _stream())
        if rc == 0:
            if ev is not None:
                TIMER.stop("igemm_wgrad", ev, 2.0 * N * IH * IW * Cin * 4 * Cout,
                           2.0 * N * IH * IW * (Cin + 4 * Cout) + 4.0 * 4 * Cin * Cout)
            _lib.call("gs_wgrad_reduce_unpack", _p(ws), parts, _p(grad), Cin, Cout, 4, 0, float(gscale), _stream())
            return
        if rc != _lib.GS_EUNSUPPORTED:
            _lib.check(rc, "gs_upconv2x2_wgrad_slabs")
    conv_wgrad_det(geom, dy, x, ws, grad, Cin, Cout, 4, gscale)


def conv_wgrad_det_batch(geoms, x, dy, ws, grad, gscale=1.0, reduce=True):
    """Deterministic weight gradients of up to four GEMMs (the sub-pixel classes of one merged transposed conv) in ONE
    launch, in the kernel layout grad[n][taps][Cout][Cin]: K parts in fp32 slabs (ws) + one ordered reduction over all
    classes; a layer that does not split K writes grad directly (ws unused).  reduce=False (K split only): the slabs
    [parts][n][taps][Cout][Cin] stay in ws for a consumer that sums them itself (upconv_split_wgrad(nparts=...)); grad may be
    None then."""
    _dev(x)
    _f32(grad, "grad")
    if x.dtype != dy.dtype:
        raise TypeError("conv_wgrad_det_batch: x and dy must share one 16-bit dtype")
    n, g0 = len(geoms), geoms[0]
    per = g0.ntaps * g0.Cout * g0.Cin
    parts = int(_lib.load().gs_conv_wgrad_parts(g0))
    if not reduce and parts <= 1:
        raise ValueError("conv_wgrad_det_batch: reduce=False needs a K split (conv_wgrad_parts > 1)")
    if reduce and (grad.numel() != n * per or not grad.is_contiguous()):
        raise ValueError("conv_wgrad_det_batch: grad must be contiguous [n][taps][Cout][Cin]")
    garr = (ctypes.POINTER(GsConvGeom) * n)(*[ctypes.pointer(g) for g in geoms])
    if parts > 1:
        _f32(ws, "ws")
        if ws.numel() < parts * n * per:
            raise ValueError("conv_wgrad_det_batch: workspace too small")
    target = grad if parts == 1 else ws
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv_wgrad_slabs_batch", n, garr, _p(x), _p(dy), _p(target), dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("igemm_wgrad", ev, sum(_geom_flops(g) for g in geoms), sum(_geom_bytes(g, True) for g in geoms))
    if parts > 1 and reduce:
        _lib.call("gs_wgrad_reduce_unpack", _p(ws), parts, _p(grad), n * g0.ntaps * g0.Cout, g0.Cin, 1, 0, float(gscale), _stream())
    elif parts == 1 and gscale != 1.0:
        grad.mul_(gscale)
    return parts


def bn_partials_numel(ntiles: int, C: int) -> int:
    return int(_lib.load().gs_bn_partials_floats(ntiles, C))


# ---------------------------------------------------------------------------- direct convs
def conv_smallcin_fwd(x, w, bias, y, bn_partials, k, stride, pad, act=ACT_NONE):
    _dev(x)
    _f32(x, "x"); _f32(w, "w"); _f32(bias, "bias"); _f32(bn_partials, "bn_partials")
    N, Cin, IH, IW = x.shape
    _, OH, OW, Cout = y.shape
    _lib.call("gs_conv_smallcin_fwd", _p(x), _p(w), _p(bias), _p(y), _p(bn_partials), N, Cin, IH, IW, Cout, OH, OW,
              k, stride, pad, act, dt_code(y), _stream())


def conv_smallcin_mtiles(N, OH, OW) -> int:
    return _lib.load().gs_conv_smallcin_mtiles(N, OH, OW)


def _direct_wgrad_ws(N, OH, OW, Cin, Cout, k, dev):
    n = int(_lib.load().gs_conv_direct_wgrad_ws_floats(N, OH, OW, Cin, Cout, k))
    return torch.empty(n, dtype=torch.float32, device=dev)


def conv_smallcin_wgrad(x, dy, dw, k, stride, pad, gscale):
    _f32(x, "x"); _f32(dw, "dw")
    N, Cin, IH, IW = x.shape
    _, OH, OW, Cout = dy.shape
    ws = _direct_wgrad_ws(N, OH, OW, Cin, Cout, k, x.device)
    _lib.call("gs_conv_smallcin_wgrad", _p(x), _p(dy), _p(dw), _p(ws), N, Cin, IH, IW, Cout, OH, OW, k, stride, pad,
              float(gscale), dt_code(dy), _stream())


def stem_bn_bwd_wgrad(y, dz, dz_stride, dz_coff, x, scale, shift, mean, invstd, c1, c2, act, dw, gscale) -> bool:
    """BatchNorm/activation backward apply + weight gradient of the one-channel 3x3 stem in one pass (dw += ...; the gradient
    w.r.t. the convolution's output stays in registers).  Returns False when the shape is outside the kernel (very wide
    images): the caller then runs bn_act_bwd_apply + conv_smallcin_wgrad."""
    _dev(y)
    _f32(x, "x"); _f32(dw, "dw")
    for n, t in (("scale", scale), ("shift", shift), ("mean", mean), ("invstd", invstd), ("c1", c1), ("c2", c2)):
        _f32(t, n)
    N, H, W, C = y.shape
    if C != 64 or tuple(x.shape) != (N, 1, H, W) or dw.numel() != 576 or y.dtype != dz.dtype:
        raise ValueError("stem_bn_bwd_wgrad: y [N,H,W,64], x [N,1,H,W], dw [64,1,3,3], one 16-bit dtype")
    if not (y.is_contiguous() and x.is_contiguous() and dw.is_contiguous()):
        raise ValueError("stem_bn_bwd_wgrad: y, x, dw must be contiguous")
    ws = _direct_wgrad_ws(N, H, W, 1, 64, 3, y.device)
    rc = _lib.load().gs_stem_bn_bwd_wgrad(_p(y), _p(dz), dz_stride, dz_coff, _p(x), _p(scale), _p(shift), _p(mean), _p(invstd),
                                          _p(c1), _p(c2), act, _p(dw), _p(ws), N, H, W, float(gscale), dt_code(y), _stream())
    if rc == _lib.GS_EUNSUPPORTED:
        return False
    if rc != 0:
        _lib.check(rc, "gs_stem_bn_bwd_wgrad")
    return True


# ---- the one-channel stem without its convolution output in memory (see include/gsseg.h)
def _stem_check(x, w, who):
    _dev(x)
    _f32(x, "x"); _f32(w, "w")
    if x.dim() != 4 or x.shape[1] != 1 or tuple(w.shape) != (64, 1, 3, 3) or not (x.is_contiguous() and w.is_contiguous()):
        raise ValueError(f"{who}: x [N,1,H,W] and w [64,1,3,3], contiguous fp32")
    return x.shape[0], x.shape[2], x.shape[3]


def stem_stats(x, w, bn_partials, tap_sums=None):
    """BatchNorm tile partials of conv(x, w) from the image alone; tap_sums [mtiles][54] fp32 (optional) receives each tile's
    nine tap sums and 45 Gram entries for stem_bwd_finalize."""
    N, H, W = _stem_check(x, w, "stem_stats")
    _f32(bn_partials, "bn_partials"); _f32(tap_sums, "tap_sums")
    mt = conv_smallcin_mtiles(N, H, W)
    if bn_partials.numel() < bn_partials_numel(mt, 64) or (tap_sums is not None and tap_sums.numel() < mt * 54):
        raise ValueError("stem_stats: bn_partials / tap_sums too small")
    _lib.call("gs_stem_stats", _p(x), _p(w), _p(bn_partials), _p(tap_sums), N, H, W, _stream())


def stem_bwd_onepass(x, z, dz, dz_stride, dz_coff, act, s1_partials, ws, z_stride=64) -> bool:
    """One pass over z (its sign = the activation's mask) and dz: s1_partials [stem_bwd_tiles][64] (sum g) and the slabs of
    A = sum g x_tap in ws [stem_bwd_tiles][576].  False: image too wide for the LDS strip.  z: dense [N,H,W,64], or (z_stride
    > 64) the first 64 channels of a wider buffer, e.g. the hi plane of a pair."""
    _dev(x)
    _f32(x, "x"); _f32(s1_partials, "s1_partials"); _f32(ws, "ws")
    N, H, W = x.shape[0], x.shape[2], x.shape[3]
    nt = stem_bwd_tiles(N, H, W)
    if (x.shape[1] != 1 or not x.is_contiguous() or s1_partials.numel() < nt * 64 or ws.numel() < nt * 576
            or tuple(z.shape) != (N, H, W, z_stride) or not z.is_contiguous() or z.dtype != dz.dtype or z_stride % 8 or z_stride < 64):
        raise ValueError("stem_bwd_onepass: x [N,1,H,W], z dense [N,H,W,z_stride], buffers of stem_bwd_tiles * 64 / 576 floats")
    rc = _lib.load().gs_stem_bwd_onepass_strided(_p(x), _p(z), z_stride, _p(dz), dz_stride, dz_coff, act, _p(s1_partials), _p(ws),
                                                 N, H, W, dt_code(z), _stream())
    if rc == _lib.GS_EUNSUPPORTED:
        return False
    if rc != 0:
        _lib.check(rc, "gs_stem_bwd_onepass")
    return True


def stem_bwd_finalize(ws, s1_partials, tap_sums, w, scale, mean, invstd, train_stats, gscale, dw, dgamma, dbeta, N, H, W):
    """BatchNorm weight / bias gradients (overwritten) and the stem weight gradient (accumulated) from the one-pass sums."""
    for n, t in (("ws", ws), ("s1_partials", s1_partials), ("tap_sums", tap_sums), ("w", w), ("dw", dw), ("dgamma", dgamma),
                 ("dbeta", dbeta)):
        _f32(t, n)
    nt = stem_bwd_tiles(N, H, W)
    if (dw.numel() != 576 or ws.numel() < nt * 576 or s1_partials.numel() < nt * 64
            or (tap_sums is not None and tap_sums.numel() < conv_smallcin_mtiles(N, H, W) * 54)):
        raise ValueError("stem_bwd_finalize: buffer sizes")
    _lib.call("gs_stem_bwd_finalize", _p(ws), _p(s1_partials), _p(tap_sums), _p(w), _p(scale), _p(mean), _p(invstd),
              int(bool(train_stats)), float(gscale), _p(dw), _p(dgamma), _p(dbeta), N, H, W, _stream())


def stem_fwd_bn(x, w, scale, shift, act, z):
    N, H, W = _stem_check(x, w, "stem_fwd_bn")
    _f32(scale, "scale"); _f32(shift, "shift")
    if tuple(z.shape) != (N, H, W, 64) or not z.is_contiguous():
        raise ValueError("stem_fwd_bn: z must be dense [N,H,W,64]")
    _lib.call("gs_stem_fwd_bn", _p(x), _p(w), _p(scale), _p(shift), act, _p(z), N, H, W, dt_code(z), _stream())


def stem_fwd_bn_pair(x, w, scale, shift, act, zpair, write_lo=True):
    """The stem in one pass into a pair buffer zpair [N,H,W,128] = [hi (64) | lo (64)]; write_lo=False leaves the lo plane alone."""
    N, H, W = _stem_check(x, w, "stem_fwd_bn_pair")
    _f32(scale, "scale"); _f32(shift, "shift")
    if tuple(zpair.shape) != (N, H, W, 128) or not zpair.is_contiguous():
        raise ValueError("stem_fwd_bn_pair: zpair must be dense [N,H,W,128]")
    lo = zpair.data_ptr() + 64 * zpair.element_size() if write_lo else None
    _lib.call("gs_stem_fwd_bn_pair", _p(x), _p(w), _p(scale), _p(shift), act, _p(zpair), lo, 128, N, H, W, dt_code(zpair), _stream())


def stem_bwd_tiles(N, H, W) -> int:
    return int(_lib.load().gs_stem_bwd_tiles(N, H, W))


def conv_smallcin_dgrad(dy, w, dx, k, stride, pad, gscale):
    _f32(w, "w"); _f32(dx, "dx")
    N, Cin, IH, IW = dx.shape
    _, OH, OW, Cout = dy.shape
    _lib.call("gs_conv_smallcin_dgrad", _p(dy), _p(w), _p(dx), N, Cin, IH, IW, Cout, OH, OW, k, stride, pad,
              float(gscale), dt_code(dy), _stream())


def conv_smallcout_fwd(x, w, bias, y, k=1, stride=1, pad=0):
    _dev(x)
    _f32(w, "w"); _f32(bias, "bias"); _f32(y, "y")
    N, IH, IW, Cin = x.shape
    _, Cout, OH, OW = y.shape
    _lib.call("gs_conv_smallcout_fwd", _p(x), _p(w), _p(bias), _p(y), N, IH, IW, Cin, Cout, OH, OW, k, stride, pad,
              dt_code(x), _stream())


def conv_smallcout_bwd(x, w, dy, dx, dw, db, k=1, stride=1, pad=0, gscale=1.0):
    _f32(w, "w"); _f32(dy, "dy"); _f32(dw, "dw"); _f32(db, "db")
    ref = x if x is not None else dx
    N, IH, IW, Cin = ref.shape
    _, Cout, OH, OW = dy.shape
    ws = _direct_wgrad_ws(N, OH, OW, Cin, Cout, k, ref.device) if dw is not None else None
    _lib.call("gs_conv_smallcout_bwd", _p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), _p(ws), N, IH, IW, Cin, Cout,
              OH, OW, k, stride, pad, float(gscale), dt_code(ref), _stream())


# ---------------------------------------------------------------------------- BatchNorm / activation
def bn_finalize(partials, ntiles, C, count, gamma, beta, running_mean, running_var, momentum, eps,
                scale, shift, mean, invstd):
    for n, t in (("gamma", gamma), ("beta", beta), ("running_mean", running_mean), ("running_var", running_var),
                 ("scale", scale), ("shift", shift), ("mean", mean), ("invstd", invstd)):
        _f32(t, n)
    _lib.call("gs_bn_finalize", _p(partials), ntiles, C, float(count), _p(gamma), _p(beta), _p(running_mean),
              _p(running_var), float(momentum), float(eps), _p(scale), _p(shift), _p(mean), _p(invstd), _stream())
    # the kernel wrote the running statistics through raw pointers: tell torch (version counters key the engines' caches)
    for t in (running_mean, running_var):
        if t is not None:
            torch.autograd.graph.increment_version(t)


def bn_eval_coeffs(C, gamma, beta, running_mean, running_var, eps, scale, shift, mean, invstd):
    _lib.call("gs_bn_eval_coeffs", C, _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps),
              _p(scale), _p(shift), _p(mean), _p(invstd), _stream())


def bn_act_apply(y, scale, shift, act, z, z_stride, z_coff, zp=None, keep_mask=None, keep_scale=1.0):
    N, H, W, C = y.shape
    if not y.is_contiguous():
        raise ValueError("bn_act_apply: y must be dense NHWC")
    _lib.call("gs_bn_act_apply", _p(y), _p(scale), _p(shift), act, _p(z), z_stride, z_coff, _p(zp), _p(keep_mask),
              float(keep_scale), N, H, W, C, dt_code(y), _stream())


def bn_bwd_tiles(N, H, W) -> int:
    return _lib.load().gs_bn_bwd_tiles(N, H, W)


def bn_bwd_tiles_used(N, H, W, pooled: bool) -> int:
    return _lib.load().gs_bn_bwd_tiles_used(N, H, W, int(pooled))


def bn_act_bwd_reduce(y, dz_a, sa, ca, dzp, scale, shift, mean, invstd, act, partials, dz_b=None, act_b=ACT_NONE,
                      keep_mask=None, keep_scale=1.0):
    N, H, W, C = y.shape
    _lib.call("gs_bn_act_bwd_reduce", _p(y), _p(dz_a), sa, ca, _p(dzp), _p(dz_b), act_b, _p(keep_mask),
              float(keep_scale), _p(scale), _p(shift), _p(mean), _p(invstd), act, _p(partials), N, H, W, C,
              dt_code(y), _stream())


def bn_bwd_coeffs(partials, ntiles, C, count, gscale, dgamma, dbeta, c1, c2):
    _lib.call("gs_bn_bwd_coeffs", _p(partials), ntiles, C, float(count), float(gscale), _p(dgamma), _p(dbeta),
              _p(c1), _p(c2), _stream())


def bn_act_bwd_apply(y, dz_a, sa, ca, dzp, scale, shift, mean, invstd, c1, c2, act, bn, dy, dz_b=None,
                     act_b=ACT_NONE, keep_mask=None, keep_scale=1.0):
    N, H, W, C = y.shape
    _lib.call("gs_bn_act_bwd_apply", _p(y), _p(dz_a), sa, ca, _p(dzp), _p(dz_b), act_b, _p(keep_mask),
              float(keep_scale), _p(scale), _p(shift), _p(mean), _p(invstd), _p(c1), _p(c2), act, int(bn), _p(dy),
              N, H, W, C, dt_code(y), _stream())


def maxpool3d_fwd(z, zp, NB, D, H, W, C, z_stride=None, z_coff=0):
    _dev(z)
    _lib.call("gs_maxpool3d_fwd", _p(z), C if z_stride is None else z_stride, z_coff, _p(zp), NB, D, H, W, C,
              dt_code(z), _stream())


def maxpool2x2_fwd(z, zp, N, H, W, C, z_stride=None, z_coff=0):
    """MaxPool2d(2) of a (strided) NHWC tensor into a dense one: the inference path's stand-alone pool."""
    _dev(z)
    if z.dtype != zp.dtype:
        raise TypeError("maxpool2x2_fwd: z and zp must share one 16-bit dtype")
    _lib.call("gs_maxpool2x2_fwd", _p(z), C if z_stride is None else z_stride, z_coff, _p(zp), N, H, W, C,
              dt_code(z), _stream())


def maxpool2x2_fwd_pair(z_hi, z_lo, z_stride, zp_hi, zp_lo, zp_stride, N, H, W, C):
    """MaxPool2d(2) on pairs (folded-BatchNorm inference of the pair forward): z_hi / z_lo two planes with pixel stride z_stride,
    pooled pair zp_hi / zp_lo (zp_lo may be None) with pixel stride zp_stride; the maximum is taken over the pair VALUES."""
    _dev(z_hi)
    if not (z_hi.dtype == z_lo.dtype == zp_hi.dtype):
        raise TypeError("maxpool2x2_fwd_pair: the planes must share one 16-bit dtype")
    _lib.call("gs_maxpool2x2_fwd_pair", _p(z_hi), _p(z_lo), z_stride, _p(zp_hi), _p(zp_lo), zp_stride, N, H, W, C, dt_code(z_hi), _stream())


def maxpool3d_bwd(z, dzp, dres, dz, NB, D, H, W, C, z_stride=None, z_coff=0, res_stride=0, res_coff=0):
    _lib.call("gs_maxpool3d_bwd", _p(z), C if z_stride is None else z_stride, z_coff, _p(dzp), _p(dres), res_stride,
              res_coff, _p(dz), NB, D, H, W, C, dt_code(z), _stream())


def head1x1_bn_fwd(y_conv, scale, shift, act, w, bias, logits):
    """1x1 head on z = act(y_conv * scale + shift) formed on the load path (the last stage's activation is never stored)."""
    _dev(y_conv)
    _f32(scale, "scale"); _f32(shift, "shift"); _f32(w, "w"); _f32(bias, "bias"); _f32(logits, "logits")
    N, H, W, C = y_conv.shape
    ncls = logits.shape[1]
    if C != 64 or tuple(logits.shape) != (N, ncls, H, W) or w.numel() != ncls * 64 or not (y_conv.is_contiguous() and logits.is_contiguous()):
        raise ValueError("head1x1_bn_fwd: y_conv [N,H,W,64], logits [N,ncls,H,W], w [ncls,64]")
    _lib.call("gs_head1x1_bn_fwd", _p(y_conv), _p(scale), _p(shift), act, _p(w), _p(bias), _p(logits), N, H, W, ncls,
              dt_code(y_conv), _stream())


def head1x1_bn_wgrad(y_conv, scale, shift, act, w, dl, dw, db, gscale=1.0):
    _dev(y_conv)
    _f32(scale, "scale"); _f32(shift, "shift"); _f32(w, "w"); _f32(dl, "dl"); _f32(dw, "dw"); _f32(db, "db")
    N, H, W, C = y_conv.shape
    ncls = dl.shape[1]
    if C != 64 or tuple(dl.shape) != (N, ncls, H, W) or dw.numel() != ncls * 64 or not (y_conv.is_contiguous() and dl.is_contiguous()):
        raise ValueError("head1x1_bn_wgrad: y_conv [N,H,W,64], dl [N,ncls,H,W], dw [ncls,64]")
    ws = _direct_wgrad_ws(N, H, W, 64, ncls, 1, y_conv.device)
    _lib.call("gs_head1x1_bn_wgrad", _p(y_conv), _p(scale), _p(shift), act, _p(w), _p(dl), _p(dw), _p(db), _p(ws), N, H, W,
              ncls, float(gscale), dt_code(y_conv), _stream())


def bn_act_bwd_reduce_head(y, dl, w_head, scale, shift, mean, invstd, act, partials):
    """bn_act_bwd_reduce with the gradient source dz = dl . w_head formed on the fly (the stage in front of a pointwise head)."""
    _dev(y)
    _f32(dl, "dl"); _f32(w_head, "w_head"); _f32(partials, "partials")
    N, H, W, C = y.shape
    ncls = dl.shape[1]
    if tuple(dl.shape) != (N, ncls, H, W) or w_head.numel() != ncls * C or not (dl.is_contiguous() and w_head.is_contiguous()):
        raise ValueError("bn_act_bwd_reduce_head: dl [N,ncls,H,W] and w_head [ncls,C] contiguous fp32")
    _lib.call("gs_bn_act_bwd_reduce_head", _p(y), _p(dl), _p(w_head), ncls, _p(scale), _p(shift), _p(mean), _p(invstd), act,
              _p(partials), N, H, W, C, dt_code(y), _stream())


def bn_act_bwd_apply_head(y, dl, w_head, scale, shift, mean, invstd, c1, c2, act, dy):
    _dev(y)
    _f32(dl, "dl"); _f32(w_head, "w_head")
    N, H, W, C = y.shape
    ncls = dl.shape[1]
    if tuple(dl.shape) != (N, ncls, H, W) or w_head.numel() != ncls * C or not (dl.is_contiguous() and w_head.is_contiguous()):
        raise ValueError("bn_act_bwd_apply_head: dl [N,ncls,H,W] and w_head [ncls,C] contiguous fp32")
    _lib.call("gs_bn_act_bwd_apply_head", _p(y), _p(dl), _p(w_head), ncls, _p(scale), _p(shift), _p(mean), _p(invstd),
              _p(c1), _p(c2), act, _p(dy), N, H, W, C, dt_code(y), _stream())


def bn_partials_colsum(partials, ntiles, Cfull, coff, C, gscale, out):
    """out[c] = gscale * sum over tiles of partials[tile][0][coff + c]: the column sums of a tensor, taken from the tile
    partials of the convolution that wrote it (ConvTranspose2d bias gradient without a pass over the tensor)."""
    _f32(partials, "partials"); _f32(out, "out")
    if partials.numel() < bn_partials_numel(ntiles, Cfull) or out.numel() < C:
        raise ValueError("bn_partials_colsum: buffer too small")
    _lib.call("gs_bn_partials_colsum", _p(partials), ntiles, Cfull, coff, C, float(gscale), _p(out), _stream())


def colsum(t, pix_stride, coff, N, H, W, y0, x0, h, w, C, gscale, ws, out):
    _f32(ws, "ws"); _f32(out, "out")
    if ws.numel() < 1024 * C:
        raise ValueError("colsum: workspace must hold 1024*C floats")
    _lib.call("gs_colsum", _p(t), pix_stride, coff, N, H, W, y0, x0, h, w, C, float(gscale), _p(ws), _p(out),
              dt_code(t), _stream())


# ---------------------------------------------------------------------------- packing / layout
def pack_weight(w, w_fwd, w_dgrad, transposed: bool):
    _dev(w)
    _f32(w, "weight")
    if transposed:
        Cin, Cout = w.shape[0], w.shape[1]
    else:
        Cout, Cin = w.shape[0], w.shape[1]
    taps = w.shape[2] * w.shape[3]
    ref = w_fwd if w_fwd is not None else w_dgrad
    _lib.call("gs_pack_weight", _p(w), _p(w_fwd), _p(w_dgrad), Cout, Cin, taps, int(transposed), dt_code(ref), _stream())


def pack_weight_multi(items):
    """items: (w, w_fwd, w_dgrad, transposed) tuples as for pack_weight (3x3 convs and k2/s2 transposed convs): one launch."""
    if not items:
        return
    descs = (_lib.GsPackDesc * len(items))()
    ref = None
    for d, (w, w_fwd, w_dgrad, transposed) in zip(descs, items):
        _dev(w)
        _f32(w, "weight")
        r = w_fwd if w_fwd is not None else w_dgrad
        if ref is None:
            ref = r
        elif r.dtype != ref.dtype:
            raise TypeError("pack_weight_multi: all packs must share one 16-bit dtype")
        cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        d.w, d.w_fwd, d.w_dgrad = _p(w), _p(w_fwd), _p(w_dgrad)
        d.Cout, d.Cin, d.taps, d.transposed = cout, cin, w.shape[2] * w.shape[3], int(transposed)
    _lib.call("gs_pack_weight_multi", len(items), descs, dt_code(ref), _stream())


def unpack_wgrad(dw, grad, A, B, taps, transposed: bool, gscale):
    _f32(dw, "dw"); _f32(grad, "grad")
    _lib.call("gs_unpack_wgrad", _p(dw), _p(grad), A, B, taps, int(transposed), float(gscale), _stream())


def nchw_to_nhwc(src, dst, dst_stride=None, dst_coff=0):
    _f32(src, "src")
    N, C, H, W = src.shape
    _lib.call("gs_nchw_to_nhwc", _p(src), _p(dst), N, C, H, W, C if dst_stride is None else dst_stride, dst_coff,
              dt_code(dst), _stream())


def nhwc_to_nchw(src, dst, src_stride=None, src_coff=0, gscale=1.0):
    _f32(dst, "dst")
    N, C, H, W = dst.shape
    _lib.call("gs_nhwc_to_nchw", _p(src), C if src_stride is None else src_stride, src_coff, _p(dst), N, C, H, W,
              float(gscale), dt_code(src), _stream())


# ---------------------------------------------------------------------------- Pix2Pix mixed up-conv
def upconv_merge_pack(w4, w6, w8, softmax3, pack_fwd=None, pack_dgrad=None, merged_f32=None):
    _dev(w8)
    for n, t in (("w4", w4), ("w6", w6), ("w8", w8), ("softmax3", softmax3), ("merged", merged_f32)):
        _f32(t, n)
    Cin, Cout = w8.shape[0], w8.shape[1]
    ref = pack_fwd if pack_fwd is not None else pack_dgrad
    code = dt_code(ref) if ref is not None else GS_F16
    _lib.call("gs_upconv_merge_pack", _p(w4), _p(w6), _p(w8), _p(softmax3), _p(pack_fwd), _p(pack_dgrad),
              _p(merged_f32), Cin, Cout, code, _stream())


def upconv8_image_fits(Cin: int, Cout: int) -> bool:
    """LDS budget of the direct outermost-layer kernel (gs_upconv8_image_fwd)."""
    return Cin % 8 == 0 and 1 <= Cout <= 4 and (144 * (Cin + 8) + 64 * Cout * Cin) * 2 <= 64 * 1024


def upconv8_image_fwd(x, pack_fwd, bias, out, u, N, h, w, Cin, Cout, act, in_stride=None, in_coff=0):
    """Merged 8x8/s2/p3 transposed conv to a 1..4-channel fp32 NCHW image (+bias, activation); u: optional 16-bit NHWC
    pre-activation [N,2h,2w,8]."""
    _dev(x)
    _f32(bias, "bias"); _f32(out, "out")
    if x.dtype != pack_fwd.dtype or (u is not None and u.dtype != x.dtype):
        raise TypeError("upconv8_image_fwd: x, pack and u must share one 16-bit dtype")
    if pack_fwd.numel() != 4 * 16 * 8 * Cin or out.numel() != N * Cout * 4 * h * w:
        raise ValueError("upconv8_image_fwd: pack / output size")
    _lib.call("gs_upconv8_image_fwd", _p(x), Cin if in_stride is None else in_stride, in_coff, _p(pack_fwd), 8, _p(bias),
              _p(out), _p(u), N, h, w, Cin, Cout, act, dt_code(x), _stream())


def upconv8_image_wgrad_ok(Cin, Cout) -> bool:
    return bool(_lib.load().gs_upconv8_image_wgrad_ok(int(Cin), int(Cout)))


def upconv8_image_wgrad(x, du, dwm, N, h, w, Cin):
    """Weight gradient of the merged 8x8/s2/p3 transposed conv to ONE image channel: x [N,h,w,>=Cin] 16-bit (channels from 0), du
    [N,2h,2w,cpad] 16-bit (channel 0), dwm fp32 [4][16][1][Cin] (un-scaled sums, the layout upconv_split_wgrad reads)."""
    _dev(x)
    _f32(dwm, "dwm")
    if x.dtype != du.dtype or x.dim() != 4 or du.dim() != 4 or tuple(x.shape[:3]) != (N, h, w) or tuple(du.shape[:3]) != (N, 2 * h, 2 * w):
        raise ValueError("upconv8_image_wgrad: x [N,h,w,C] and du [N,2h,2w,cpad] of one 16-bit dtype")
    if not (x.is_contiguous() and du.is_contiguous()) or dwm.numel() != 64 * Cin or not dwm.is_contiguous():
        raise ValueError("upconv8_image_wgrad: dense x / du, dwm [4][16][1][Cin]")
    ws = torch.empty(int(_lib.load().gs_upconv8_image_wgrad_ws_floats(N, h, w, Cin)), dtype=torch.float32, device=x.device)
    _lib.call("gs_upconv8_image_wgrad", _p(x), x.shape[3], _p(du), du.shape[3], _p(ws), _p(dwm), N, h, w, Cin, dt_code(x), _stream())


def upconv_split_wgrad_parts_ok(Cin, Cout) -> bool:
    return bool(_lib.load().gs_upconv_split_wgrad_parts_ok(int(Cin), int(Cout)))


def upconv_split_wgrad(dwm, w4, w6, w8, softmax3, gscale, dw4, dw6, dw8, dots3, nparts=1):
    """merged weight gradient [4][16][Cout][Cin] -> dW4 / dW6 / dW8 (+ the three architecture dot products).  nparts > 1: dwm is
    the stack of split-K slabs of conv_wgrad_det_batch(reduce=False), summed in part order on the way."""
    for n, t in (("dwm", dwm), ("dw4", dw4), ("dw6", dw6), ("dw8", dw8), ("dots3", dots3)):
        _f32(t, n)
    Cin, Cout = w8.shape[0], w8.shape[1]
    # deterministic form: per-block partial dot products in a scratch tensor, summed in block order
    ws = torch.empty(int(_lib.load().gs_upconv_split_wgrad_ws_floats(Cin, Cout)), dtype=torch.float32, device=dwm.device)
    if nparts > 1:
        if dwm.numel() < nparts * 64 * Cin * Cout:
            raise ValueError("upconv_split_wgrad: the slab stack is too small")
        _lib.call("gs_upconv_split_wgrad_parts", _p(dwm), int(nparts), 64 * Cin * Cout, _p(w4), _p(w6), _p(w8), _p(softmax3),
                  float(gscale), _p(dw4), _p(dw6), _p(dw8), _p(dots3), _p(ws), Cin, Cout, _stream())
        return
    _lib.call("gs_upconv_split_wgrad_det", _p(dwm), _p(w4), _p(w6), _p(w8), _p(softmax3), float(gscale), _p(dw4), _p(dw6),
              _p(dw8), _p(dots3), _p(ws), Cin, Cout, _stream())


def fake_postprocess(x, out, gamma_lut):
    """gs_fake_postprocess: min-max -> uint8 -> equalise -> gamma -> float of an fp32 batch [N,C,H,W]; every
    (image, channel) plane has its own histogram (torchvision equalises channel by channel), min/max are global."""
    _dev(x)
    _f32(x, "x"); _f32(out, "out"); _f32(gamma_lut, "gamma_lut")
    if not x.is_contiguous() or out.shape != x.shape or gamma_lut.numel() != 256:
        raise ValueError("fake_postprocess: contiguous x, out of the same shape and a 256-entry gamma table")
    N = x.shape[0] * (x.shape[1] if x.dim() == 4 else 1)
    hw = x.numel() // N
    ws = torch.empty(int(_lib.load().gs_fake_postprocess_ws_floats(N)), dtype=torch.float32, device=x.device)
    _lib.call("gs_fake_postprocess", _p(x), _p(out), _p(ws), _p(gamma_lut), N, hw, _stream())


def isic_fake_trans(x, out, equalize_on, bits, sharpness_on, sharpness_factor, autocontrast_on, saturation_factor):
    """gs_isic_fake_trans on an fp32 batch [N,C,H,W]; saturation_factor None = no saturation step.  The blend ratios are
    passed as (float32(r), float32(1.0 - r)) with the subtraction in double, as torchvision's `_blend` evaluates them."""
    _dev(x)
    _f32(x, "x"); _f32(out, "out")
    if x.dim() != 4 or not x.is_contiguous() or out.shape != x.shape or not out.is_contiguous():
        raise ValueError("isic_fake_trans: contiguous x, out [N,C,H,W] of one shape")
    N, C, H, W = x.shape
    ws = torch.empty(int(_lib.load().gs_isic_fake_trans_ws_bytes(N * C, H * W)), dtype=torch.uint8, device=x.device)
    sat_on = saturation_factor is not None and C == 3
    sr = float(saturation_factor) if sat_on else 1.0
    _lib.call("gs_isic_fake_trans", _p(x), _p(out), _p(ws), N, C, H, W, int(bool(equalize_on)), int(bits),
              int(bool(sharpness_on)), float(sharpness_factor), float(1.0 - float(sharpness_factor)), int(bool(autocontrast_on)),
              int(sat_on), sr, float(1.0 - sr), _stream())


# ---------------------------------------------------------------------------- losses
LOSS_WS = 4 * 1024


def seg_loss_fwd(logits, mask_u8, ws, out):
    _dev(logits)
    _f32(logits, "logits"); _f32(ws, "ws"); _f32(out, "out")
    if mask_u8.dtype != torch.uint8 or not mask_u8.is_contiguous():
        raise TypeError("mask must be contiguous uint8 [N,H,W]")
    N, C, H, W = logits.shape
    if mask_u8.numel() != N * H * W:
        raise ValueError("mask / logits shape mismatch")
    _lib.call("gs_seg_loss_fwd", _p(logits), _p(mask_u8), N, C, H, W, _p(ws), _p(out), _stream())


def seg_loss_bwd(logits, mask_u8, out, gout, gscale, dlogits):
    N, C, H, W = logits.shape
    _f32(dlogits, "dlogits"); _f32(gout, "gout")
    _lib.call("gs_seg_loss_bwd", _p(logits), _p(mask_u8), _p(out), _p(gout), float(gscale), _p(dlogits), N, C, H, W,
              _stream())


def dice_loss_fwd(p, t, ws, out):
    _dev(p)
    _f32(p, "input"); _f32(t, "target")
    _lib.call("gs_dice_loss_fwd", _p(p), _p(t), p.numel(), _p(ws), _p(out), _stream())


def dice_loss_bwd(t, out, gout, dp):
    _lib.call("gs_dice_loss_bwd", _p(t), _p(out), _p(gout), _p(dp), t.numel(), _stream())


def dice_coeff_batched(p, t, out):
    """Per-item Dice coefficients of p, t [B, n_per] fp32 in one launch pair: out[0] = mean, out[1 + b] = dice_b."""
    _dev(p)
    _f32(p, "p"); _f32(t, "t"); _f32(out, "out")
    B, n_per = p.shape
    if t.shape != p.shape or out.numel() < 1 + B:
        raise ValueError("dice_coeff_batched: p, t must be [B, n] of one shape and out hold 1 + B floats")
    ws = torch.empty(int(_lib.load().gs_dice_batched_ws_floats(B)), dtype=torch.float32, device=p.device)
    _lib.call("gs_dice_coeff_batched", _p(p), _p(t), B, n_per, _p(ws), _p(out), _stream())


def eval_dice(logits, mask_u8, out):
    """Fused evaluation head (unet/evaluate.py:29-43): threshold / arg-max + per-(sample, foreground class) Dice, mean in out[0]."""
    _dev(logits)
    _f32(logits, "logits"); _f32(out, "out")
    N, C, H, W = logits.shape
    if mask_u8.dtype != torch.uint8 or not mask_u8.is_contiguous() or mask_u8.numel() != N * H * W:
        raise TypeError("eval_dice: mask must be contiguous uint8 [N,H,W]")
    B = N * max(1, C - 1)
    if out.numel() < 1 + B:
        raise ValueError("eval_dice: out must hold 1 + N*max(1,C-1) floats")
    ws = torch.empty(int(_lib.load().gs_dice_batched_ws_floats(B)), dtype=torch.float32, device=logits.device)
    _lib.call("gs_eval_dice", _p(logits), _p(mask_u8), N, C, H * W, _p(ws), _p(out), _stream())


def jaccard_seg_loss_fwd(logits, mask_u8, out):
    _dev(logits)
    _f32(logits, "logits"); _f32(out, "out")
    N, C, H, W = logits.shape
    if C != 1:
        raise ValueError("the Jaccard loss of train_end2end_isic.py is the one-class form")
    if mask_u8.dtype != torch.uint8 or not mask_u8.is_contiguous() or mask_u8.numel() != N * H * W:
        raise TypeError("mask must be contiguous uint8 [N,H,W]")
    if out.numel() < int(_lib.load().gs_jaccard_loss_out_floats(N)):
        raise ValueError("jaccard_seg_loss_fwd: out too small")
    ws = torch.empty(int(_lib.load().gs_dice_batched_ws_floats(N)), dtype=torch.float32, device=logits.device)
    _lib.call("gs_jaccard_seg_loss_fwd", _p(logits), _p(mask_u8), N, H * W, _p(ws), _p(out), _stream())


def jaccard_seg_loss_bwd(logits, mask_u8, out, gout, gscale, dlogits):
    _f32(gout, "gout"); _f32(dlogits, "dlogits")
    N, C, H, W = logits.shape
    _lib.call("gs_jaccard_seg_loss_bwd", _p(logits), _p(mask_u8), _p(out), _p(gout), float(gscale), _p(dlogits), N, H * W,
              _stream())


def eval_jaccard(logits, mask_u8, out):
    """Fused ISIC validation metric (train_end2end_isic.py:58-84): threshold + per-sample Jaccard index, mean in out[0]."""
    _dev(logits)
    _f32(logits, "logits"); _f32(out, "out")
    N, C, H, W = logits.shape
    if C != 1 or mask_u8.dtype != torch.uint8 or not mask_u8.is_contiguous() or mask_u8.numel() != N * H * W or out.numel() < 1 + N:
        raise ValueError("eval_jaccard: one-class logits, contiguous uint8 mask [N,H,W], out of 1 + N floats")
    ws = torch.empty(int(_lib.load().gs_dice_batched_ws_floats(N)), dtype=torch.float32, device=logits.device)
    _lib.call("gs_eval_jaccard", _p(logits), _p(mask_u8), N, H * W, _p(ws), _p(out), _stream())


def mean_loss_fwd(x, t, cval, mode, ws, out):
    _dev(x)
    _f32(x, "x"); _f32(t, "target")
    _lib.call("gs_mean_loss_fwd", _p(x), _p(t), float(cval), mode, x.numel(), _p(ws), _p(out), _stream())


def mean_loss_bwd(x, t, cval, mode, gout, gscale, dx):
    _lib.call("gs_mean_loss_bwd", _p(x), _p(t), float(cval), mode, x.numel(), _p(gout), float(gscale), _p(dx), _stream())


# ---------------------------------------------------------------------------- precise mode (hi/lo 16-bit pairs)
# Tensors named *_hi / *_lo are the two planes of a pair; they may be channel slices of one buffer (the conv kernels
# want [hi | lo] side by side: in_wrap = channels of both planes) or two dense tensors.  Strides are in elements.
def pack_weight_split(w, pack, transposed: bool):
    """fp32 conv weight [Cout][Cin][kh][kw] (transposed: [Cin][Cout][kh][kw]) -> pack [taps][Cout][3*Cin] = [hi | hi | lo]."""
    _dev(w)
    _f32(w, "w")
    taps = w.shape[2] * w.shape[3]
    cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    if pack.numel() != taps * cout * 3 * cin or not pack.is_contiguous():
        raise ValueError("pack_weight_split: pack must be contiguous [taps][Cout][3*Cin]")
    _lib.call("gs_pack_weight_split", _p(w), _p(pack), cout, cin, taps, 1 if transposed else 0, dt_code(pack), _stream())


def conv3x3_precise(x, w, y_hi, y_lo, N, H, W, Cin, Cout, in_stride, in_coff=0, out_stride=None, out_coff=0,
                    bias=None, bn_partials=None, act=ACT_NONE, taps=TAPS3_FWD):
    """3x3/s1/p1 convolution on pairs: x holds the [hi | lo] planes of Cin channels each (2*Cin consecutive channels at
    in_coff), w = pack_weight_split pack [9][Cout][3*Cin]; the result leaves as y_hi / y_lo (same out_stride / out_coff)."""
    _dev(x)
    _f32(bias, "bias"); _f32(bn_partials, "bn_partials")
    if not (x.dtype == w.dtype == y_hi.dtype == y_lo.dtype):
        raise TypeError("conv3x3_precise: x, w, y must share one 16-bit dtype")
    if Cin % 64 != 0 or w.numel() != 9 * Cout * 3 * Cin:
        raise ValueError("conv3x3_precise: Cin must be a multiple of 64 and w the [9][Cout][3*Cin] split pack")
    if bn_partials is not None and bn_partials.numel() < bn_partials_numel(conv3x3_stat_rows(N, H, W, Cin, Cout, pair=True), Cout):
        raise ValueError("conv3x3_precise: bn_partials too small")
    dy, dx = _tap_arrays(taps)
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv3x3_precise", _p(x), _p(w), _p(y_hi), _p(y_lo), _p(bias), _p(bn_partials), N, H, W, 3 * Cin,
              in_stride, in_coff, 2 * Cin, Cout, Cout if out_stride is None else out_stride, out_coff, dy, dx, act,
              dt_code(x), _stream())
    if ev is not None:                       # algorithmic work of the convolution (the kernel executes 3x the MFMAs)
        TIMER.stop("conv3x3_halo_precise", ev, 2.0 * N * H * W * Cout * 9 * Cin,
                   2.0 * (N * H * W * 2 * (Cin + Cout) + 9 * 3 * Cin * Cout))


def upconv2x2_fwd_precise(x, w, bias, y_hi, y_lo, N, IH, IW, Cin, Cout, OH, OW, in_stride, in_coff=0, out_stride=None,
                          out_coff=0, ooy=0, oox=0):
    """ConvTranspose2d(k 2, s 2) on pairs: x = [hi | lo] planes (2*Cin channels at in_coff), w = pack_weight_split(transposed)
    [4][Cout][3*Cin]; output pair y_hi / y_lo (same stride / offset), e.g. the up half of both planes of a concat buffer."""
    _dev(x)
    _f32(bias, "bias")
    if not (x.dtype == w.dtype == y_hi.dtype == y_lo.dtype):
        raise TypeError("upconv2x2_fwd_precise: x, w, y must share one 16-bit dtype")
    if w.numel() != 4 * Cout * 3 * Cin:
        raise ValueError("upconv2x2_fwd_precise: w must be the [4][Cout][3*Cin] split pack")
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_upconv2x2_fwd_precise", _p(x), _p(w), _p(bias), _p(y_hi), _p(y_lo), N, IH, IW, 3 * Cin, in_stride,
              in_coff, 2 * Cin, Cout, OH, OW, Cout if out_stride is None else out_stride, out_coff, ooy, oox,
              dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("igemm_fwd_precise", ev, 2.0 * N * IH * IW * Cin * 4 * Cout,
                   2.0 * (N * IH * IW * 2 * (Cin + 4 * Cout) + 4 * K * Cout))


def pack_weight_segs(items):
    """Segment packs for the mixed-precision pair forward, all in ONE launch.  items: (w, pack, transposed, segs) with
    segs = [(kind, ci0, len), ...] (kind 0 = hi(w), 1 = lo(w), 2 = zeros); pack = contiguous [taps][Cout][sum len] 16-bit."""
    if not items:
        return
    descs = (_lib.GsSegPackDesc * len(items))()
    ref = items[0][1]
    for d, (w, pack, transposed, segs) in zip(descs, items):
        _dev(w)
        _f32(w, "weight")
        if pack.dtype != ref.dtype:
            raise TypeError("pack_weight_segs: all packs must share one 16-bit dtype")
        cin, cout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
        taps = w.shape[2] * w.shape[3]
        if not 1 <= len(segs) <= _lib.GS_SEG_MAX:
            raise ValueError("pack_weight_segs: 1..4 segments")
        ktot = sum(s_[2] for s_ in segs)
        if pack.numel() != taps * cout * ktot or not pack.is_contiguous() or not w.is_contiguous():
            raise ValueError("pack_weight_segs: pack must be contiguous [taps][Cout][sum of segment lengths]")
        d.w, d.pack = _p(w), _p(pack)
        d.Cout, d.Cin, d.taps, d.transposed, d.nseg = cout, cin, taps, int(transposed), len(segs)
        for j, (kind, ci0, ln) in enumerate(segs):
            d.kind[j], d.ci0[j], d.len[j] = int(kind), int(ci0), int(ln)
    _lib.call("gs_pack_weight_segs", len(items), descs, dt_code(ref), _stream())


def conv3x3_segs(x, w, y_hi, y_lo, N, H, W, K, wrap, Cin, Cout, in_stride, in_coff=0, out_stride=None, out_coff=0,
                 bias=None, bn_partials=None, act=ACT_NONE, taps=TAPS3_FWD):
    """3x3/s1/p1 convolution of the pair forward with a free choice of MFMA segments: the K extent (a multiple of 64,
    wrap <= K <= 2*wrap) runs over the input channels [in_coff, in_coff + wrap) of x and wraps to in_coff once; w is the
    matching pack_weight_segs pack [9][Cout][K]; Cin = the layer's channels (for the FLOP count only).  Result: pair
    y_hi / y_lo."""
    _dev(x)
    _f32(bias, "bias"); _f32(bn_partials, "bn_partials")
    if not (x.dtype == w.dtype == y_hi.dtype == y_lo.dtype):
        raise TypeError("conv3x3_segs: x, w, y must share one 16-bit dtype")
    if K % 64 != 0 or wrap % 64 != 0 or not (wrap <= K <= 2 * wrap) or w.numel() != 9 * Cout * K:
        raise ValueError("conv3x3_segs: K / wrap must be multiples of 64 with wrap <= K <= 2*wrap and w the [9][Cout][K] pack")
    if bn_partials is not None and bn_partials.numel() < bn_partials_numel(conv3x3_stat_rows(N, H, W, Cin, Cout, pair=True), Cout):
        raise ValueError("conv3x3_segs: bn_partials too small (conv3x3_stat_rows(pair=True) rows of [2][Cout])")
    dy, dx = _tap_arrays(taps)
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv3x3_precise", _p(x), _p(w), _p(y_hi), _p(y_lo), _p(bias), _p(bn_partials), N, H, W, K,
              in_stride, in_coff, wrap, Cout, Cout if out_stride is None else out_stride, out_coff, dy, dx, act,
              dt_code(x), _stream())
    if ev is not None:                       # algorithmic work of the convolution (the kernel executes K / Cin times the MFMAs)
        TIMER.stop("conv3x3_halo_precise", ev, 2.0 * N * H * W * Cout * 9 * Cin,
                   2.0 * (N * H * W * 2 * (Cin + Cout) + 9 * K * Cout))


def conv3d3_segs(x, w, y_hi, y_lo, NB, D, H, W, K, wrap, Cin, Cout, in_stride, in_coff=0, bn_partials=None, wrap_to=0):
    """Conv3d(k3, p1) of the pair forward (UNet3D): conv3x3_segs with depth -- x holds `wrap` channels per voxel ([hi | lo] planes or
    a part of them), K (a multiple of 64, wrap <= K <= 2*wrap) runs over them and wraps once, w = pack_weight_segs pack
    [27][Cout][K]; Cin = the layer's channels (FLOP count only).  wrap_to: the wrapped part of K continues at this input channel (a
    multiple of 64) instead of channel 0.  Result: dense pair y_hi / y_lo [NB*D, H, W, Cout]."""
    _dev(x)
    _f32(bn_partials, "bn_partials")
    if not (x.dtype == w.dtype == y_hi.dtype == y_lo.dtype):
        raise TypeError("conv3d3_segs: x, w, y must share one 16-bit dtype")
    if K % 64 != 0 or wrap % 64 != 0 or not (wrap <= K <= 2 * wrap) or w.numel() != 27 * Cout * K:
        raise ValueError("conv3d3_segs: K / wrap must be multiples of 64 with wrap <= K <= 2*wrap and w the [27][Cout][K] pack")
    if bn_partials is not None and bn_partials.numel() < bn_partials_numel(conv3x3_stat_rows(NB * D, H, W, K, Cout, pair=True), Cout):
        raise ValueError("conv3d3_segs: bn_partials too small (conv3x3_stat_rows(NB*D, H, W, K, Cout, pair=True) rows)")
    dz, dy, dx = _TAPS3D[1]
    ev = TIMER.start() if TIMER is not None else None
    if wrap_to:
        if wrap_to % 64 != 0 or wrap_to + (K - wrap) > wrap:
            raise ValueError("conv3d3_segs: wrap_to must be a multiple of 64 with wrap_to + K - wrap <= wrap")
        _lib.call("gs_conv3d_3x3x3_precise_to", _p(x), _p(w), _p(y_hi), _p(y_lo), None, _p(bn_partials), NB, D, H, W, K, in_stride,
                  in_coff, wrap, wrap_to, Cout, Cout, 0, dz, dy, dx, ACT_NONE, dt_code(x), _stream())
    else:
        _lib.call("gs_conv3d_3x3x3_precise", _p(x), _p(w), _p(y_hi), _p(y_lo), None, _p(bn_partials), NB, D, H, W, K, in_stride,
                  in_coff, wrap, Cout, Cout, 0, dz, dy, dx, ACT_NONE, dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("conv3x3_halo_precise", ev, 2.0 * NB * D * H * W * Cout * 27 * Cin,
                   2.0 * (NB * D * H * W * 2 * (Cin + Cout) + 27 * K * Cout))


def maxpool3d_fwd_pair(z_hi, z_lo, z_stride, zp_hi, zp_lo, zp_stride, NB, D, H, W, C):
    """MaxPool3d(2) on a pair: z_hi / z_lo are views whose first element is the first channel of the plane (pixel stride z_stride);
    zp_hi / zp_lo (None: not stored) likewise with zp_stride."""
    _lib.call("gs_maxpool3d_fwd_pair", _p(z_hi), _p(z_lo), z_stride, _p(zp_hi), _p(zp_lo), zp_stride, NB, D, H, W, C,
              dt_code(z_hi), _stream())


def upsample2x_bilinear_fwd_pair(x_hi, x_lo, y_hi, y_lo, N, IH, IW, C, OH, OW, in_stride, out_stride, out_coff=0, ooy=0, oox=0):
    """bilinear x2 (align_corners=True) of a pair: views x_hi / x_lo (pixel stride in_stride) -> y_hi / y_lo (None: not stored;
    pixel stride out_stride, channel offset out_coff inside the [N,OH,OW,*] buffers, pad offset (ooy, oox))."""
    _dev(x_hi)
    _lib.call("gs_upsample2x_bilinear_fwd_pair", _p(x_hi), _p(x_lo), _p(y_hi), _p(y_lo), N, IH, IW, C, in_stride, 0, OH, OW,
              out_stride, out_coff, ooy, oox, dt_code(x_hi), _stream())


def upconv2x2_fwd_segs(x, w, bias, y_hi, y_lo, N, IH, IW, K, wrap, Cin, Cout, OH, OW, in_stride, in_coff=0, out_stride=None,
                       out_coff=0, ooy=0, oox=0):
    """ConvTranspose2d(k 2, s 2) of the pair forward with a free choice of segments (see conv3x3_segs); w = [4][Cout][K]."""
    _dev(x)
    _f32(bias, "bias")
    if not (x.dtype == w.dtype == y_hi.dtype == y_lo.dtype):
        raise TypeError("upconv2x2_fwd_segs: x, w, y must share one 16-bit dtype")
    if w.numel() != 4 * Cout * K or not (wrap <= K <= 2 * wrap):
        raise ValueError("upconv2x2_fwd_segs: w must be the [4][Cout][K] pack, wrap <= K <= 2*wrap")
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_upconv2x2_fwd_precise", _p(x), _p(w), _p(bias), _p(y_hi), _p(y_lo), N, IH, IW, K, in_stride,
              in_coff, wrap, Cout, OH, OW, Cout if out_stride is None else out_stride, out_coff, ooy, oox,
              dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("igemm_fwd_precise", ev, 2.0 * N * IH * IW * Cin * 4 * Cout,
                   2.0 * (N * IH * IW * 2 * (Cin + 4 * Cout) + 4 * K * Cout))


def conv_smallcin_fwd_split(x, w, y_hi, y_lo, bn_partials, k, pad):
    """First conv (fp32 NCHW image, fp32 weights, stride 1) -> dense pair y_hi / y_lo [N,H,W,Cout]."""
    _dev(x)
    _f32(x, "x"); _f32(w, "w"); _f32(bn_partials, "bn_partials")
    N, Cin, H, W = x.shape
    Cout = y_hi.shape[3]
    if tuple(y_hi.shape) != (N, H, W, Cout) or y_lo.shape != y_hi.shape or not (y_hi.is_contiguous() and y_lo.is_contiguous()):
        raise ValueError("conv_smallcin_fwd_split: y_hi / y_lo must be dense [N,H,W,Cout]")
    _lib.call("gs_conv_smallcin_fwd_split", _p(x), _p(w), _p(y_hi), _p(y_lo), _p(bn_partials), N, Cin, H, W, Cout, k, pad,
              dt_code(y_hi), _stream())


def bn_act_apply_split(y_hi, y_lo, scale, shift, act, z_hi, z_lo, z_stride, z_coff, zp_hi=None, zp_lo=None, zp_stride=0):
    """z pair = act(scale * (y_hi + y_lo) + shift); y_hi / y_lo dense [N,H,W,C]; optional 2x2 max-pooled pair."""
    N, H, W, C = y_hi.shape
    if not (y_hi.is_contiguous() and y_lo.is_contiguous()) or y_lo.shape != y_hi.shape:
        raise ValueError("bn_act_apply_split: y_hi / y_lo must be dense NHWC of one shape")
    _f32(scale, "scale"); _f32(shift, "shift")
    _lib.call("gs_bn_act_apply_split", _p(y_hi), _p(y_lo), _p(scale), _p(shift), act, _p(z_hi), _p(z_lo), z_stride, z_coff,
              _p(zp_hi), _p(zp_lo), zp_stride, N, H, W, C, dt_code(y_hi), _stream())


def bn_act_apply_split_pool3d(y_hi, y_lo, scale, shift, act, z_hi, z_lo, z_stride, z_coff, zp_hi, zp_lo, zp_stride, NB, D, H, W):
    """bn_act_apply_split + MaxPool3d(2) of the pair in one pass: y_hi / y_lo dense [NB*D,H,W,C]; z pair as bn_act_apply_split; the
    pooled pair zp_hi / zp_lo (views of the first channel of each plane, pixel stride zp_stride) = the maximum of the stored pair values."""
    C = y_hi.shape[3]
    if tuple(y_hi.shape) != (NB * D, H, W, C) or y_lo.shape != y_hi.shape or not (y_hi.is_contiguous() and y_lo.is_contiguous()):
        raise ValueError("bn_act_apply_split_pool3d: y_hi / y_lo must be dense [NB*D,H,W,C]")
    _f32(scale, "scale"); _f32(shift, "shift")
    _lib.call("gs_bn_act_apply_split_pool3d", _p(y_hi), _p(y_lo), _p(scale), _p(shift), act, _p(z_hi), _p(z_lo), z_stride, z_coff,
              _p(zp_hi), _p(zp_lo), zp_stride, NB, D, H, W, C, dt_code(y_hi), _stream())


def head1x1_bn_fwd_split(y_hi, y_lo, scale, shift, act, w, bias, logits):
    """OutConv 1x1 on act(scale * (y_hi + y_lo) + shift) of a dense conv-output pair [N,H,W,64] -> fp32 NCHW logits."""
    _dev(y_hi)
    _f32(scale, "scale"); _f32(shift, "shift"); _f32(w, "w"); _f32(bias, "bias"); _f32(logits, "logits")
    N, H, W, C = y_hi.shape
    ncls = logits.shape[1]
    if (C != 64 or y_lo.shape != y_hi.shape or tuple(logits.shape) != (N, ncls, H, W) or w.numel() != ncls * 64
            or not (y_hi.is_contiguous() and y_lo.is_contiguous() and logits.is_contiguous())):
        raise ValueError("head1x1_bn_fwd_split: y pair [N,H,W,64] dense, logits [N,ncls,H,W], w [ncls,64]")
    _lib.call("gs_head1x1_bn_fwd_split", _p(y_hi), _p(y_lo), _p(scale), _p(shift), act, _p(w), _p(bias), _p(logits), N, H, W, 64,
              ncls, dt_code(y_hi), _stream())


def head1x1_fwd_split(x_hi, x_lo, w, bias, y):
    """OutConv 1x1 on a dense pair [N,H,W,64] -> fp32 NCHW logits."""
    _dev(x_hi)
    _f32(w, "w"); _f32(bias, "bias"); _f32(y, "y")
    N, H, W, Cin = x_hi.shape
    if not (x_hi.is_contiguous() and x_lo.is_contiguous()) or x_lo.shape != x_hi.shape:
        raise ValueError("head1x1_fwd_split: x_hi / x_lo must be dense NHWC of one shape")
    _lib.call("gs_head1x1_fwd_split", _p(x_hi), _p(x_lo), _p(w), _p(bias), _p(y), N, H, W, Cin, y.shape[1], dt_code(x_hi),
              _stream())


# ---------------------------------------------------------------------------- "q" stages: FP8 correction segment of the pair forward
# (include/gsseg.h, csrc/common.hpp): the lo plane of a pair travels as a Q PLANE -- per 32 channels 64 bytes [lo8 | hi8] of e4m3 --
# and a conv stage runs x_hi.w_hi on the 16-bit MFMA plus ONE block-scaled e4m3 segment for x_lo.w_hi + x_hi.w_lo.
def conv3x3_q8_ok(W, Cin, Cout) -> bool:
    """does the "q" form of a 3x3 / 3x3x3 conv exist for this shape (LDS-DMA kernel: W >= 24, Cin % 64 == 0, Cout % 8 == 0)?"""
    return USE_HALO_CONV and bool(_lib.load().gs_conv3x3_q8_ok(W, Cin, Cout))


def pack_weight_q8(items):
    """items: (w fp32 [Cout][Cin][taps...], pack [taps][Cout][2*Cin] 16-bit elements = 4*Cin bytes per row, wexp int32 [Cout]); one launch."""
    if not items:
        return
    descs = (_lib.GsQ8PackDesc * len(items))()
    ref = items[0][1]
    for d, (w, pack, wexp) in zip(descs, items):
        _dev(w)
        _f32(w, "weight")
        cout, cin = w.shape[0], w.shape[1]
        taps = w.numel() // (cout * cin)
        if pack.numel() != taps * cout * 2 * cin or not pack.is_contiguous() or pack.dtype != ref.dtype:
            raise ValueError("pack_weight_q8: pack must be contiguous [taps][Cout][2*Cin] of the engine's 16-bit dtype")
        if wexp.dtype != torch.int32 or wexp.numel() != cout or not wexp.is_contiguous():
            raise ValueError("pack_weight_q8: wexp must be int32 [Cout]")
        d.w, d.pack, d.wexp, d.Cout, d.Cin, d.taps = _p(w), _p(pack), _p(wexp), cout, cin, taps
    _lib.call("gs_pack_weight_q8", len(items), descs, dt_code(ref), _stream())


def conv3x3_q8(x, w, wexp, y_hi, y_lo, N, H, W, Cin, Cout, in_stride, in_coff=0, bn_partials=None):
    """3x3/s1/p1 conv as a "q" stage: x = [hi plane (Cin) | q plane] at in_coff (pixel stride in_stride), w / wexp from pack_weight_q8;
    result: dense pair y_hi / y_lo [N,H,W,Cout]."""
    _dev(x)
    _f32(bn_partials, "bn_partials")
    if not (x.dtype == w.dtype == y_hi.dtype == y_lo.dtype == torch.float16):
        raise TypeError("conv3x3_q8: fp16 tensors only")
    if w.numel() != 9 * Cout * 2 * Cin:
        raise ValueError("conv3x3_q8: w must be the [9][Cout][2*Cin] q pack")
    if bn_partials is not None and bn_partials.numel() < bn_partials_numel(conv3x3_stat_rows(N, H, W, 2 * Cin, Cout, pair="q"), Cout):
        raise ValueError("conv3x3_q8: bn_partials too small")
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv3x3_q8", _p(x), _p(w), _p(wexp), _p(y_hi), _p(y_lo), _p(bn_partials), N, H, W, Cin, in_stride, in_coff,
              Cout, Cout, 0, dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("conv3x3_halo_precise", ev, 2.0 * N * H * W * Cout * 9 * Cin,
                   2.0 * (N * H * W * 2 * (Cin + Cout) + 9 * 2 * Cin * Cout))


def conv3d3_q8(x, w, wexp, y_hi, y_lo, NB, D, H, W, Cin, Cout, in_stride, in_coff=0, bn_partials=None):
    """Conv3d(k3, p1) as a "q" stage (see conv3x3_q8); w = [27][Cout][2*Cin] q pack."""
    _dev(x)
    _f32(bn_partials, "bn_partials")
    if not (x.dtype == w.dtype == y_hi.dtype == y_lo.dtype == torch.float16):
        raise TypeError("conv3d3_q8: fp16 tensors only")
    if w.numel() != 27 * Cout * 2 * Cin:
        raise ValueError("conv3d3_q8: w must be the [27][Cout][2*Cin] q pack")
    if bn_partials is not None and bn_partials.numel() < bn_partials_numel(conv3x3_stat_rows(NB * D, H, W, 2 * Cin, Cout, pair="q"), Cout):
        raise ValueError("conv3d3_q8: bn_partials too small")
    ev = TIMER.start() if TIMER is not None else None
    _lib.call("gs_conv3d_3x3x3_q8", _p(x), _p(w), _p(wexp), _p(y_hi), _p(y_lo), _p(bn_partials), NB, D, H, W, Cin, in_stride,
              in_coff, Cout, Cout, 0, dt_code(x), _stream())
    if ev is not None:
        TIMER.stop("conv3x3_halo_precise", ev, 2.0 * NB * D * H * W * Cout * 27 * Cin,
                   2.0 * (NB * D * H * W * 2 * (Cin + Cout) + 27 * 2 * Cin * Cout))


def bn_act_apply_split_q8(y_hi, y_lo, scale, shift, act, z_hi, z_lo, z_q8, z_stride, z_coff, zp_hi=None, zp_lo=None, zp_q8=False,
                          zp_stride=0):
    """bn_act_apply_split with q planes: z_q8 / zp_q8 say whether z_lo / zp_lo are q planes (pointers at the plane's byte 0; z_coff
    selects the chunk) or 16-bit lo planes."""
    N, H, W, C = y_hi.shape
    if not (y_hi.is_contiguous() and y_lo.is_contiguous()) or y_lo.shape != y_hi.shape:
        raise ValueError("bn_act_apply_split_q8: y_hi / y_lo must be dense NHWC of one shape")
    _f32(scale, "scale"); _f32(shift, "shift")
    _lib.call("gs_bn_act_apply_split_q8", _p(y_hi), _p(y_lo), _p(scale), _p(shift), act, _p(z_hi), _p(z_lo), int(bool(z_q8)), z_stride,
              z_coff, _p(zp_hi), _p(zp_lo), int(bool(zp_q8)), zp_stride, N, H, W, C, dt_code(y_hi), _stream())


def stem_fwd_bn_pair_q8(x, w, scale, shift, act, zpair):
    """stem_fwd_bn_pair writing [hi plane (64) | q plane] into zpair [N,H,W,128]."""
    _stem_check(x, w, "stem_fwd_bn_pair_q8")
    _f32(scale, "scale"); _f32(shift, "shift")
    N, _, H, W = x.shape
    if tuple(zpair.shape) != (N, H, W, 128) or not zpair.is_contiguous():
        raise ValueError("stem_fwd_bn_pair_q8: zpair must be dense [N,H,W,128]")
    _lib.call("gs_stem_fwd_bn_pair_q8", _p(x), _p(w), _p(scale), _p(shift), act, _p(zpair), _p(zpair[..., 64:]), 128, N, H, W,
              dt_code(zpair), _stream())


def q8_from_hi(buf, q_plane, pixels, C, pix_stride, coff):
    """q plane (hi8 from the stored hi plane, lo8 = 0) of channels [coff, coff + C) of a buffer whose hi plane starts at `buf` and
    whose q plane starts at `q_plane` (both with pixel stride pix_stride 16-bit elements)."""
    _dev(buf)
    _lib.call("gs_q8_from_hi", _p(buf), _p(q_plane), int(pixels), C, pix_stride, coff, dt_code(buf), _stream())


def maxpool3d_fwd_pair_q8(z_hi, z_lo, z_q8, zq_coff, z_stride, zp_hi, zp_lo, zp_q8, zp_stride, NB, D, H, W, C):
    """maxpool3d_fwd_pair with q planes: z_q8 -> z_lo is the input buffer's q plane (byte 0), the pooled channels start at its channel
    zq_coff; zp_q8 -> zp_lo is the pooled buffer's q plane."""
    _lib.call("gs_maxpool3d_fwd_pair_q8", _p(z_hi), _p(z_lo), int(bool(z_q8)), zq_coff, z_stride, _p(zp_hi), _p(zp_lo), int(bool(zp_q8)),
              zp_stride, NB, D, H, W, C, dt_code(z_hi), _stream())
