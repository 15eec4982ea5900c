"""Tensor-level operators: PyTorch-ROCm tensors in HBM -> libparesis_hip.so (ctypes) on the current HIP stream.

PyTorch is plumbing here (device memory, streams, torch.distributed); all arithmetic on the hot path runs in the
hand-written HIP kernels behind the C ABI.  Every function checks devices/dtypes/shapes on the host before a kernel is
launched, and raises PsxError on any failure -- there is no CPU fallback.
"""
import ctypes
from ctypes import c_double, c_float, c_int, c_void_p

import torch

from . import _lib
from ._lib import PsxError, check, lib

MARGIN_FRESNEL = 15   # Experiment.py:236
MARGIN_DETECTOR = 15  # Detector.py:92


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(None)


def _need(t, dtype, name, shape=None):
    if not isinstance(t, torch.Tensor):
        raise PsxError("%s must be a torch.Tensor, got %r" % (name, type(t)))
    if not t.is_cuda:
        raise PsxError("%s must live in HBM (cuda/ROCm tensor); got device %s. There is no CPU path." % (name, t.device))
    if t.dtype != dtype:
        raise PsxError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise PsxError("%s must be contiguous (row-major [Nx][Ny])" % name)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise PsxError("%s has shape %s, expected %s" % (name, tuple(t.shape), tuple(shape)))
    return t


class MaterialStack:
    """Thickness maps [nmat, Nx, Ny] float32 in HBM + per-map coefficients (see include/paresis_hip.h, "Materials")."""

    def __init__(self, T, cphase=None, catt=None):
        if T is None:
            self.T, self.n = None, 0
        else:
            _need(T, torch.float32, "thickness stack")
            if T.dim() != 3:
                raise PsxError("Sample Geometry has the wrong nb of dim [material, x, y]")   # Sample.py:263-264
            self.T, self.n = T, T.shape[0]
        if self.n > _lib.PSX_MAX_MAT:
            raise PsxError("at most %d materials per call, got %d" % (_lib.PSX_MAX_MAT, self.n))
        self.cphase = [0.0] * self.n if cphase is None else [float(v) for v in cphase]
        self.catt = [0.0] * self.n if catt is None else [float(v) for v in catt]
        if len(self.cphase) != self.n or len(self.catt) != self.n:
            raise PsxError("coefficient lists must have one entry per material")

    @staticmethod
    def concat(*stacks):
        """Materials of several objects seen by one kernel (e.g. membrane phase + sample phase, Experiment.py:469)."""
        stacks = [s for s in stacks if s is not None and s.n > 0]
        if not stacks:
            return MaterialStack(None)
        out = MaterialStack.__new__(MaterialStack)
        out.T = None
        out.n = sum(s.n for s in stacks)
        if out.n > _lib.PSX_MAX_MAT:
            raise PsxError("at most %d materials per call, got %d" % (_lib.PSX_MAX_MAT, out.n))
        out.cphase = [c for s in stacks for c in s.cphase]
        out.catt = [c for s in stacks for c in s.catt]
        out._maps = [s.map(i) for s in stacks for i in range(s.n)]
        return out

    def map(self, i):
        if self.T is not None:
            return self.T[i]
        return self._maps[i]

    def with_coeffs(self, cphase=None, catt=None):
        out = MaterialStack.__new__(MaterialStack)
        out.T, out.n = self.T, self.n
        if self.T is None and self.n:
            out._maps = self._maps
        out.cphase = list(self.cphase) if cphase is None else [float(v) for v in cphase]
        out.catt = list(self.catt) if catt is None else [float(v) for v in catt]
        return out

    def cargs(self, shape=None):
        """(T** host array, cphase*, catt*, nmat) for the C ABI; keeps the ctypes arrays alive on self."""
        maps = [self.map(i) for i in range(self.n)]
        for i, m in enumerate(maps):
            _need(m, torch.float32, "thickness map %d" % i, shape)
        self._c = ((c_void_p * max(1, self.n))(*[m.data_ptr() for m in maps]),
                   (c_double * max(1, self.n))(*self.cphase), (c_double * max(1, self.n))(*self.catt))
        return self._c[0], self._c[1], self._c[2], self.n


_NO_MATS = None


def _mats(m):
    global _NO_MATS
    if m is None:
        if _NO_MATS is None:
            _NO_MATS = MaterialStack(None)
        return _NO_MATS
    return m


class MaterialBatch:
    """The materials of a batch of sources over the same maps (the energies of a detector bin): `base` carries the maps,
    cphase[s][i] / catt[s][i] the coefficients of source s.  What a list of per-source MaterialStack objects says, without
    building them (and re-checking their maps) for every membrane position: the coefficients depend on the energy only."""

    def __init__(self, base, cphase, catt):
        self.base = _mats(base)
        self.cphase = [[float(v) for v in r] for r in cphase]
        self.catt = [[float(v) for v in r] for r in catt]
        if any(len(r) != self.base.n for r in self.cphase) or any(len(r) != self.base.n for r in self.catt):
            raise PsxError("MaterialBatch: one coefficient per material and source")
        self.n = self.base.n

    def rebase(self, base):
        """The same coefficients over another position's maps."""
        out = MaterialBatch.__new__(MaterialBatch)
        out.base, out.cphase, out.catt, out.n = _mats(base), self.cphase, self.catt, self.n
        if out.base.n != self.n:
            raise PsxError("MaterialBatch.rebase: %d maps for %d coefficients" % (out.base.n, self.n))
        return out


def _batch_mats(mats, ns, shape, what):
    """(T** host array, nmat, cphase[ns][nmat], catt[ns][nmat]) from a MaterialBatch or from one MaterialStack per source."""
    if isinstance(mats, MaterialBatch):
        if len(mats.cphase) != ns:
            raise PsxError("%s: %d coefficient rows for %d sources" % (what, len(mats.cphase), ns))
        return mats.base.cargs(shape)[0], mats.n, mats.cphase, mats.catt
    mats = [_mats(None)] * ns if mats is None else [_mats(m) for m in mats]
    nm = mats[0].n
    T = mats[0].cargs(shape)[0]
    for m in mats[1:]:
        if m.n != nm or any(m.map(i).data_ptr() != mats[0].map(i).data_ptr() for i in range(nm)):
            raise PsxError("%s: every source must use the same thickness maps" % what)
    return T, nm, [m.cphase for m in mats], [m.catt for m in mats]


_workspaces = {}
_status_words = {}


def _workspace(device, nbytes):
    # one scratch buffer per (device, stream): calls issued on different streams may overlap on the GPU
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def status_word(device):
    """Per-device status word the kernels OR error bits into (PSX_STATUS_*)."""
    w = _status_words.get(device.index)
    if w is None:
        w = torch.zeros(1, dtype=torch.int32, device=device)
        _status_words[device.index] = w
    return w


def check_status(device, what="refraction"):
    """Synchronising read of the status word; raises like refractionFileNumba2.py:81-82 and clears it."""
    w = status_word(device)
    v = int(w.item())
    if v:
        w.zero_()
        if v & _lib.STATUS_NONFINITE:
            raise PsxError("The calculated intensity refractive includes some nans or insane values (%s)" % what)
        raise PsxError("device status %d after %s" % (v, what))


# --------------------------------------------------------------------------------------------- transmission
def transmit_wave(wave_in, amp, mats, out=None):
    """K1 (Sample.py:279): out = amp * wave_in * exp(sum catt*T) * exp(i sum cphase*T).  wave_in None = unit wave."""
    mats = _mats(mats)
    ref = wave_in if wave_in is not None else mats.map(0)
    shape = tuple(ref.shape)
    if wave_in is not None:
        _need(wave_in, torch.complex64, "wave_in")
    if out is None:
        out = torch.empty(shape, dtype=torch.complex64, device=ref.device)
    _need(out, torch.complex64, "wave_out", shape)
    T, cp, ca, n = mats.cargs(shape)
    check(lib().psx_transmit_wave_c64(_ptr(wave_in), c_float(amp), T, cp, ca, n, _ptr(out), out.numel(), _stream()),
          "psx_transmit_wave_c64")
    return out


def fill(t, value):
    """t[:] = value through the library's own elementwise kernel (K2 with no input image and no material): the hot loop
    launches no PyTorch kernel -- the first launch of one (fill, cat, reduce) makes PyTorch load its code object for that
    kernel family, a one-off 15-40 ms host stall that showed up at membrane position 0 (position loop)."""
    _need(t, torch.float32, "fill target")
    check(lib().psx_transmit_rt_f32(None, c_float(value), None, None, None, 0, _ptr(t), None, None, t.numel(), _stream()),
          "psx_transmit_rt_f32 (fill)")
    return t


def transmit_rt(I_in, I0, mats, phi_in=None, want_phi=True, shape=None, out=None):
    """K2 (Sample.py:347-348): I = I0*I_in*exp(sum catt*T); phi = phi_in + sum cphase*T (float64)."""
    mats = _mats(mats)
    ref = I_in if I_in is not None else (mats.map(0) if mats.n else out)
    shape = tuple(ref.shape)
    if I_in is not None:
        _need(I_in, torch.float32, "I_in")
    if phi_in is not None:
        _need(phi_in, torch.float64, "phi_in", shape)
    if out is not None:
        _need(out, torch.float32, "I_out", shape)
    I_out = out if out is not None else torch.empty(shape, dtype=torch.float32, device=ref.device)
    phi_out = torch.empty(shape, dtype=torch.float64, device=ref.device) if want_phi else None
    T, cp, ca, n = mats.cargs(shape)
    check(lib().psx_transmit_rt_f32(_ptr(I_in), c_float(I0), T, cp, ca, n, _ptr(I_out), _ptr(phi_in), _ptr(phi_out),
                                    I_out.numel(), _stream()), "psx_transmit_rt_f32")
    return I_out, phi_out


def accumulate(acc, img, scale=1.0, mats=None, add=True):
    """acc (+)= scale*img*exp(sum catt*T)  (Experiment.py:351-358, 478-483)."""
    mats = _mats(mats)
    _need(acc, torch.float32, "acc")
    _need(img, torch.float32, "img", acc.shape)
    T, cp, ca, n = mats.cargs(tuple(acc.shape))
    check(lib().psx_accumulate_f32(_ptr(acc), _ptr(img), c_float(scale), T, ca, n, 1 if add else 0, acc.numel(),
                                   _stream()), "psx_accumulate_f32")
    return acc


def new_sums(device):
    return torch.zeros((_lib.PSX_SUM_SLOTS, _lib.PSX_SUM_STRIDE), dtype=torch.float64, device=device)


def fold_sums(sums):
    """[sum S, sum weight*S] (float64 tensor of 2, still in HBM) of a new_sums() buffer."""
    return sums[:, :2].sum(dim=0)


def accumulate_sum(acc, img, sums, weight, scale=1.0, mats=None, add=True):
    """accumulate() that also reduces what it adds: sums[0] += sum(v), sums[1] += weight*sum(v), v = scale*img*att
    (Experiment.py:360-361, 485-486).  sums: float64 [SUM_SLOTS, SUM_STRIDE] in HBM from new_sums() -- the kernel's
    workgroups add into 32 slots 128 bytes apart; fold_sums() gives the two totals.  acc may be None (reduction only)."""
    mats = _mats(mats)
    _need(img, torch.float32, "img")
    if acc is not None:
        _need(acc, torch.float32, "acc", img.shape)
    _need(sums, torch.float64, "sums", (_lib.PSX_SUM_SLOTS, _lib.PSX_SUM_STRIDE))
    T, cp, ca, n = mats.cargs(tuple(img.shape))
    check(lib().psx_accumulate_sum_f32(_ptr(acc), _ptr(img), c_float(scale), T, ca, n, 1 if add else 0, img.numel(),
                                       _ptr(sums), c_double(weight), _stream()), "psx_accumulate_sum_f32")
    return acc


def accumulate_many(acc, imgs, sums, weights, scales=None, mats=None, add=True):
    """accumulate_sum() over the images of several energies in ONE pass per PSX_MAX_SRC images, added in list order exactly
    as one accumulate_sum call per image would: acc (+)= sum_e scales[e] * imgs[e] * att_e, sums += (sum, sum of
    weights[e] * term).  mats[e]: MaterialStack of image e (same maps, own attenuation coefficients) or None."""
    ne = len(imgs)
    scales = [1.0] * ne if scales is None else [float(v) for v in scales]
    shape = tuple(imgs[0].shape)
    for i, t in enumerate(imgs):
        _need(t, torch.float32, "imgs[%d]" % i, shape)
    if acc is not None:
        _need(acc, torch.float32, "acc", shape)
    if sums is not None:
        _need(sums, torch.float64, "sums", (_lib.PSX_SUM_SLOTS, _lib.PSX_SUM_STRIDE))
    T, nm, _, cat = _batch_mats(mats, ne, shape, "accumulate_many")
    for e0 in range(0, ne, _lib.PSX_MAX_SRC):
        el = range(e0, min(ne, e0 + _lib.PSX_MAX_SRC))
        n = len(el)
        check(lib().psx_accumulate_many_f32(
            _ptr(acc), (c_void_p * n)(*[imgs[e].data_ptr() for e in el]), (c_float * n)(*[scales[e] for e in el]), n, T,
            (c_double * max(1, n * nm))(*[c for e in el for c in cat[e]]), nm, 1 if (add or e0 > 0) else 0,
            imgs[0].numel(), _ptr(sums), (c_double * n)(*[float(weights[e]) for e in el]), _stream()),
            "psx_accumulate_many_f32")
    return acc


# ----------------------------------------------------------------------------------------------- refraction
def refract(shape, mats, dscale, clamp, margin=15, I_in=None, I0=1.0, phi_in=None, out=None, out_scale=1.0, add=False,
            want_D=False, I_mut=None):
    """K9-K13 (refractionFileNumba2.py:25-86 + 198-263).  Returns (I_out, Dx|None, Dy|None); Dx,Dy are padded."""
    mats = _mats(mats)
    Nx, Ny = int(shape[0]), int(shape[1])
    dev = (I_in if I_in is not None else (phi_in if phi_in is not None else mats.map(0))).device
    if I_in is not None:
        _need(I_in, torch.float32, "I_in", (Nx, Ny))
    if phi_in is not None:
        _need(phi_in, torch.float64, "phi_in", (Nx, Ny))
    if out is None:
        if add:
            raise PsxError("add=True needs an existing output image")
        out = torch.empty((Nx, Ny), dtype=torch.float32, device=dev)
    _need(out, torch.float32, "I_out", (Nx, Ny))
    Dx = Dy = None
    if want_D:
        Dx = torch.empty((Nx + 2 * margin, Ny + 2 * margin), dtype=torch.float32, device=dev)
        Dy = torch.empty_like(Dx)
    if I_mut is not None and (I_in is None or I_mut.data_ptr() != I_in.data_ptr()):
        raise PsxError("I_mut must alias I_in")
    ws = _workspace(dev, lib().psx_refract_workspace_bytes(Nx, Ny))
    T, cp, ca, n = mats.cargs((Nx, Ny))
    check(lib().psx_refract_f32(_ptr(I_in), c_float(I0), T, cp, ca, n, _ptr(phi_in), _ptr(out), c_float(out_scale),
                                1 if add else 0, _ptr(Dx), _ptr(Dy), _ptr(I_mut), Nx, Ny, int(margin), c_double(dscale),
                                c_double(clamp[0]), c_double(clamp[1]), _ptr(status_word(dev)), _ptr(ws), _stream()),
          "psx_refract_f32")
    return out, Dx, Dy


def refract_split(shape, mats, dscale, clamp, mask, margin=15, I_in=None, I0=1.0, phi_in=None, outs=None, out_scale=1.0,
                  add=False):
    """fastRefractionDF's split and its two refractions (RF2:147-154) in one call (psx_refract_split_f32): returns
    (refraction of the sources where mask == 0, refraction of the sources where mask != 0); mask: float32 [Nx][Ny]."""
    mats = _mats(mats)
    Nx, Ny = int(shape[0]), int(shape[1])
    _need(mask, torch.float32, "mask", (Nx, Ny))
    dev = mask.device
    if I_in is not None:
        _need(I_in, torch.float32, "I_in", (Nx, Ny))
    if phi_in is not None:
        _need(phi_in, torch.float64, "phi_in", (Nx, Ny))
    if outs is None:
        if add:
            raise PsxError("add=True needs existing output images")
        outs = [torch.empty((Nx, Ny), dtype=torch.float32, device=dev) for _ in range(2)]
    for o in outs:
        _need(o, torch.float32, "I_out", (Nx, Ny))
    ws = _workspace(dev, lib().psx_refract_multi_workspace_bytes(Nx, Ny, 2))
    T, cp, ca, n = mats.cargs((Nx, Ny))
    check(lib().psx_refract_split_f32(_ptr(I_in), _ptr(mask), c_float(I0), T, cp, ca, n, _ptr(phi_in), _ptr(outs[0]),
                                      _ptr(outs[1]), c_float(out_scale), 1 if add else 0, Nx, Ny, int(margin), c_double(dscale),
                                      c_double(clamp[0]), c_double(clamp[1]), _ptr(status_word(dev)), _ptr(ws), _stream()),
          "psx_refract_split_f32")
    return outs[0], outs[1]


def refract_batch(shape, mats, dscales, clamp, margin=15, I_in=None, I0=None, outs=None, out_scale=1.0, add=False):
    """Several refractions over the SAME thickness maps in one launch per kernel (the energies of a detector bin,
    EXP:448-486): mats[e] (same maps, own coefficients), dscales[e], I_in[e] (all or None) or the uniform I0[e], outs[e].
    Every image is what refract() gives for that refraction.  Returns the list of output images."""
    ne = len(dscales)
    Nx, Ny = int(shape[0]), int(shape[1])
    T, nm, cph, cat = _batch_mats(mats, ne, (Nx, Ny), "refract_batch")
    dev = (mats.base if isinstance(mats, MaterialBatch) else _mats(mats[0])).map(0).device
    if I_in is not None:
        for e, t in enumerate(I_in):
            _need(t, torch.float32, "I_in[%d]" % e, (Nx, Ny))
    if outs is None:
        if add:
            raise PsxError("add=True needs existing output images")
        outs = [torch.empty((Nx, Ny), dtype=torch.float32, device=dev) for _ in range(ne)]
    for e, t in enumerate(outs):
        _need(t, torch.float32, "outs[%d]" % e, (Nx, Ny))
    I0 = [1.0] * ne if I0 is None else [float(v) for v in I0]
    for e0 in range(0, ne, _lib.PSX_MAX_SRC):
        el = range(e0, min(ne, e0 + _lib.PSX_MAX_SRC))
        n = len(el)
        ws = _workspace(dev, lib().psx_refract_batch_workspace_bytes(Nx, Ny, n))
        check(lib().psx_refract_batch_f32(
            n, (c_void_p * n)(*[I_in[e].data_ptr() for e in el]) if I_in is not None else None,
            (c_float * n)(*[I0[e] for e in el]), T,
            (c_double * (n * nm))(*[c for e in el for c in cph[e]]),
            (c_double * (n * nm))(*[c for e in el for c in cat[e]]), nm,
            (c_void_p * n)(*[outs[e].data_ptr() for e in el]), c_float(out_scale), 1 if add else 0, Nx, Ny, int(margin),
            (c_double * n)(*[float(dscales[e]) for e in el]), c_double(clamp[0]), c_double(clamp[1]),
            _ptr(status_word(dev)), _ptr(ws), _stream()), "psx_refract_batch_f32")
    return outs


def refract_multi(shape, mats, dscales, clamp, margin=15, I_in=None, I0=1.0, phi_in=None, outs=None, out_scale=1.0,
                  add=False):
    """A propagation-distance batch of K9-K13: len(dscales) refractions of the SAME source (I_in/I0, mats, phi_in) in
    one launch per kernel (psx_refract_multi_f32); the thickness maps are read once per tile for all distances.
    Returns the list of images, each identical to refract(..., dscale=dscales[d])[0]."""
    mats = _mats(mats)
    Nx, Ny = int(shape[0]), int(shape[1])
    nd = len(dscales)
    dev = (I_in if I_in is not None else (phi_in if phi_in is not None else mats.map(0))).device
    if I_in is not None:
        _need(I_in, torch.float32, "I_in", (Nx, Ny))
    if phi_in is not None:
        _need(phi_in, torch.float64, "phi_in", (Nx, Ny))
    if outs is None:
        if add:
            raise PsxError("add=True needs existing output images")
        outs = [torch.empty((Nx, Ny), dtype=torch.float32, device=dev) for _ in range(nd)]
    if len(outs) != nd:
        raise PsxError("refract_multi: %d output images for %d distances" % (len(outs), nd))
    for o in outs:
        _need(o, torch.float32, "I_out", (Nx, Ny))
    ws = _workspace(dev, lib().psx_refract_multi_workspace_bytes(Nx, Ny, nd))
    T, cp, ca, n = mats.cargs((Nx, Ny))
    optr = (c_void_p * nd)(*[o.data_ptr() for o in outs])
    dsc = (c_double * nd)(*[float(x) for x in dscales])
    check(lib().psx_refract_multi_f32(_ptr(I_in), c_float(I0), T, cp, ca, n, _ptr(phi_in), optr, c_float(out_scale),
                                      1 if add else 0, None, None, None, Nx, Ny, int(margin), dsc, nd,
                                      c_double(clamp[0]), c_double(clamp[1]), _ptr(status_word(dev)), _ptr(ws),
                                      _stream()), "psx_refract_multi_f32")
    return outs


def set_refract_halo(halo):
    """Gather halo of the refraction tile kernel: 4, 6, 8, 12 or 16 pixels (psx_refract_set_halo; a speed knob -- the halo decides
    which shares are gathered in the tiles and which are replayed, so the last bit of an image may depend on it)."""
    check(lib().psx_refract_set_halo(int(halo)), "psx_refract_set_halo")


def tune_refract_halo(call, halos=(4, 6, 8), reps=2):
    """Picks the gather halo by MEASUREMENT: `call()` (a refraction with the caller's real arguments) is timed with each halo --
    one untimed run, then `reps` runs between two events -- and the fastest stays set.  Which halo wins depends on how many
    rays travel further than it (the far-ray replay costs ~11 global atomics per ns): 4 or 6 pixels at oversampling 2, 12 at
    oversampling 4 with the bench's membranes (pass halos=(4, 6, 8, 12, 16) there).  The images agree to float rounding whatever the choice (a pixel's sum is split
    differently between tile gather and replay: the last bit may move -- a reproducible run fixes the halo by rule instead).
    One host synchronisation per candidate: meant for the set-up of a run, not for its loop.  Returns (halo, {halo: ms})."""
    times = {}
    for h in halos:
        set_refract_halo(h)
        call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        e1.synchronize()
        times[h] = e0.elapsed_time(e1) / reps
    best = min(times, key=times.get)
    set_refract_halo(best)
    return best, times


def darkfield_blur(I2DF, DF, I2, R):
    """Variable-width Gaussian re-splat of fastRefractionDF (refractionFileNumba2.py:168-186): returns
    I2 + sum_s I2DF[s] * gaussian_shape(DF[s]/2) centred on s.  DF in pixels at the target pixels; R = max half-size."""
    _need(I2DF, torch.float32, "I2DF")
    _need(DF, torch.float32, "DF", I2DF.shape)
    if I2 is not None:
        _need(I2, torch.float32, "I2", I2DF.shape)
    out = torch.empty_like(I2DF)
    Nx, Ny = I2DF.shape
    ws = torch.empty(lib().psx_darkfield_workspace_bytes(Nx, Ny), dtype=torch.uint8, device=I2DF.device)
    check(lib().psx_darkfield_blur_f32(_ptr(I2DF), _ptr(DF), _ptr(I2), _ptr(out), Nx, Ny, int(R), _ptr(ws), _stream()),
          "psx_darkfield_blur_f32")
    return out


def darkfield_split(I, DF_rad, num, den, limit):
    """The front of fastRefractionDF in one pass (psx_darkfield_split_f32); DF_px = DF_rad * num / den, float64, in the
    reference's order (RF2:114: num = propagationDistance, den = studyPixelSize*1e-6*magnification).  Returns (I_nodf, I_df, DF_px float32, prep,
    words): words = two device uint64 holding the float64 bit patterns of max(DF_px) before / after the DF > limit -> 0 rule
    -- read them with darkfield_maxima() only when the caller does not know the maximum (one host synchronisation)."""
    _need(I, torch.float32, "I")
    _need(DF_rad, torch.float64, "darkField", I.shape)
    Nx, Ny = I.shape
    dev = I.device
    I_nodf, I_df, DF_px = (torch.empty((Nx, Ny), dtype=torch.float32, device=dev) for _ in range(3))
    prep = torch.empty(lib().psx_darkfield_workspace_bytes(Nx, Ny), dtype=torch.uint8, device=dev)
    words = torch.empty(2, dtype=torch.int64, device=dev)
    check(lib().psx_darkfield_split_f32(_ptr(I), _ptr(DF_rad), c_double(num), c_double(den), c_double(limit), _ptr(I_nodf), _ptr(I_df),
                                        _ptr(DF_px), _ptr(prep), _ptr(words), Nx, Ny, _stream()), "psx_darkfield_split_f32")
    return I_nodf, I_df, DF_px, prep, words


def darkfield_maxima(words):
    """(max DF_px, max DF_px after the DF > limit rule) from the device words of darkfield_split: ONE device-to-host copy."""
    import numpy as np
    w = words.cpu().numpy().view(np.float64)
    return float(w[0]), float(w[1])


def darkfield_blur_prepared(I2DF, DF, prep, I2, R, scan=True, out=None, add=False):
    """darkfield_blur() on the patch table darkfield_split() made from the width map; scan: NaN / inf in the result raise
    the device status word (RF2:190-193), checked by the kernel that stores it.  out / add: the result is stored into (added
    to) an existing image -- the chain's sum over energies without a pass of its own."""
    _need(I2DF, torch.float32, "I2DF")
    _need(DF, torch.float32, "DF", I2DF.shape)
    if out is None:
        if add:
            raise PsxError("add=True needs an existing output image")
        out = torch.empty_like(I2DF)
    _need(out, torch.float32, "out", I2DF.shape)
    Nx, Ny = I2DF.shape
    check(lib().psx_darkfield_blur_prepared_f32(_ptr(I2DF), _ptr(DF), _ptr(prep), _ptr(I2), _ptr(out), Nx, Ny, int(R),
                                                _ptr(status_word(I2DF.device)) if scan else None, 1 if add else 0, _stream()),
          "psx_darkfield_blur_prepared_f32")
    return out


def darkfield_merge(I, a, b):
    """I[:] = a + b through the library (no PyTorch kernel in the dark-field path)."""
    check(lib().psx_darkfield_merge_f32(_ptr(I), _ptr(a), _ptr(b), I.numel(), _stream()), "psx_darkfield_merge_f32")
    return I


def repad(src, margin_src, margin_dst, shape):
    """The centre `shape` of a map padded by margin_src, re-padded with zeros to margin_dst (psx_repad_f32)."""
    Nx, Ny = int(shape[0]), int(shape[1])
    _need(src, torch.float32, "src", (Nx + 2 * margin_src, Ny + 2 * margin_src))
    dst = torch.empty((Nx + 2 * margin_dst, Ny + 2 * margin_dst), dtype=torch.float32, device=src.device)
    check(lib().psx_repad_f32(_ptr(src), int(margin_src), _ptr(dst), int(margin_dst), Nx, Ny, _stream()), "psx_repad_f32")
    return dst


def set_deterministic(on=True):
    """Order-independent sums in the scatter paths that otherwise use float atomics (far rays, fastloop): fixed-point
    deposits, bitwise reproducible results, a few per cent slower.  Per host thread; off by default in the
    library, ON around the Experiment class's ray-tracing chain unless exp_dict['reproducible'] is False."""
    check(lib().psx_set_deterministic(1 if on else 0), "psx_set_deterministic")


def get_deterministic():
    return bool(lib().psx_get_deterministic())


def set_deterministic_scale(scale=0.0):
    """The replay's fixed-point unit from the caller's intensity scale (psx_set_deterministic_scale); 0: from the call's own
    measured maximum (the default)."""
    check(lib().psx_set_deterministic_scale(c_float(float(scale))), "psx_set_deterministic_scale")


def clock_probe():
    """Shader clock in MHz the device sustains right now (psx_clock_probe: a 30 us spin on every CU behind what is queued on the
    current stream, which it synchronises).  Diagnostics: the boxes of a pool differ by > 10 % in what they sustain."""
    import ctypes
    mhz = ctypes.c_float(0.0)
    check(lib().psx_clock_probe(ctypes.byref(mhz), _stream()), "psx_clock_probe")
    return float(mhz.value)


def get_deterministic_scale():
    """The calling thread's replay scale (psx_get_deterministic_scale; 0 = the unit comes from each call's measured maximum)."""
    return float(lib().psx_get_deterministic_scale())


class deterministic:
    """with ops.deterministic(on[, scale]): ... -- sets the mode (and the unit's scale) and puts back what the caller had, mode AND
    scale: scopes nest (Experiment.refraction inside the ray-tracing chain's scope keeps the chain's fixed unit; a user's own
    psx_set_deterministic_scale survives a call of the class).  scale=None keeps the caller's scale."""

    def __init__(self, on=True, scale=0.0):
        self.on = bool(on)
        self.scale = None if scale is None else (float(scale) if on else 0.0)

    def __enter__(self):
        self.prev, self.prev_scale = get_deterministic(), get_deterministic_scale()
        set_deterministic(self.on)
        if self.scale is not None:
            set_deterministic_scale(self.scale)
        return self

    def __exit__(self, *exc):
        set_deterministic(self.prev)
        set_deterministic_scale(self.prev_scale)
        return False


def debug_switch(name, value=1):
    """A diagnostic A/B switch of the library (psx_debug_switch; include/paresis_hip.h lists the names).  Process-wide, off by
    default; for tools and tests -- the library never reads the environment."""
    check(lib().psx_debug_switch(str(name).encode(), int(value)), "psx_debug_switch")


def debug_switches_active():
    """'name=value ...' of every diagnostic switch that is not at its default ('' = a clean product run)."""
    buf = ctypes.create_string_buffer(512)
    check(lib().psx_debug_switches_active(buf, len(buf)), "psx_debug_switches_active")
    return buf.value.decode()


def fastloop(I, Dx, Dy, I2):
    """fastloopNumba (refractionFileNumba2.py:198-263) on explicit float32 displacement maps; accumulates into I2."""
    _need(I, torch.float32, "I")
    for nm, t in (("Dx", Dx), ("Dy", Dy), ("I2", I2)):
        _need(t, torch.float32, nm, I.shape)
    check(lib().psx_fastloop_f32(_ptr(I), _ptr(Dx), _ptr(Dy), _ptr(I2), I.shape[0], I.shape[1], _stream()),
          "psx_fastloop_f32")
    return I2


# -------------------------------------------------------------------------------------------------- Fresnel
class FresnelPlan:
    """psx_fresnel_plan for one study grid (Experiment.wavePropagation, Experiment.py:219-252)."""

    def __init__(self, Nx, Ny, margin=MARGIN_FRESNEL, max_dist=4, engine=_lib.ENGINE_AUTO, device=None):
        self.Nx, self.Ny, self.margin, self.max_dist = int(Nx), int(Ny), int(margin), int(max_dist)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._h = c_void_p(None)
        with torch.cuda.device(self.device):
            check(lib().psx_fresnel_plan_create(self.Nx, self.Ny, self.margin, self.max_dist, int(engine),
                                                ctypes.byref(self._h)), "psx_fresnel_plan_create")
        self.engine = lib().psx_fresnel_plan_engine(self._h)
        self.bytes = lib().psx_fresnel_plan_bytes(self._h)

    def work_queue(self, on=True):
        """Line groups through a queue instead of static shares (psx_fresnel_plan_work_queue): for runs whose transfers or
        other streams share the GPU with the propagations."""
        check(lib().psx_fresnel_plan_work_queue(self._h, 1 if on else 0), "psx_fresnel_plan_work_queue")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().psx_fresnel_plan_destroy(self._h)
            self._h = c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def propagate(self, a, gphase, du, wave_in=None, amp=1.0, mats=None, want_wave=None, inten_out=None,
                  inten_scale=None, add=False):
        """One input wave -> len(a) distances.  a[d]=z/(2kM), gphase[d]=kz/M, du=(du_x,du_y)=2*pi/(N*h).

        want_wave[d]: return the complex field; inten_out[d]: float32 image that receives inten_scale[d]*|.|^2
        (added when add=True).  Returns the list of complex outputs (None where not wanted)."""
        mats = _mats(mats)
        nd = len(a)
        shape = (self.Nx, self.Ny)
        if wave_in is not None:
            _need(wave_in, torch.complex64, "wave_in", shape)
        want_wave = [True] * nd if want_wave is None else list(want_wave)
        inten_out = [None] * nd if inten_out is None else list(inten_out)
        inten_scale = [1.0] * nd if inten_scale is None else [float(s) for s in inten_scale]
        waves = [torch.empty(shape, dtype=torch.complex64, device=self.device) if w else None for w in want_wave]
        for i, t in enumerate(inten_out):
            if t is not None:
                _need(t, torch.float32, "inten_out[%d]" % i, shape)
        T, cp, ca, n = mats.cargs(shape)
        wo = (c_void_p * nd)(*[w.data_ptr() if w is not None else None for w in waves])
        io = (c_void_p * nd)(*[t.data_ptr() if t is not None else None for t in inten_out])
        check(lib().psx_fresnel_propagate(self._h, _ptr(wave_in), c_float(amp), T, cp, ca, n, nd,
                                          (c_double * nd)(*[float(v) for v in a]),
                                          (c_double * nd)(*[float(v) for v in gphase]), c_double(du[0]), c_double(du[1]),
                                          wo, io, (c_float * nd)(*inten_scale), 1 if add else 0, _stream()),
              "psx_fresnel_propagate")
        return waves


    def propagate_sources(self, a, gphase, du, wave_in=None, amp=None, mats=None, want_wave=None, inten_out=None,
                          inten_scale=None):
        """Several source waves over the SAME thickness maps in one call (the energies of a detector bin, EXP:317-361):
        a[s][d], gphase[s][d]; wave_in[s] (or None), amp[s]; mats[s]: MaterialStack of source s (same maps, own coefficients);
        want_wave[d]; inten_out[s][d] (float32 image or None), inten_scale[s][d].  Every (source, distance) result is what
        propagate() gives for that source; nothing is accumulated.  Returns waves[s][d] (None where not wanted).  More than
        PSX_MAX_SRC sources are taken in chunks."""
        ns, nd = len(a), len(a[0])
        shape = (self.Nx, self.Ny)
        T, nm, cph, cat = _batch_mats(mats, ns, shape, "propagate_sources")
        amp = [1.0] * ns if amp is None else [float(v) for v in amp]
        want_wave = [True] * nd if want_wave is None else list(want_wave)
        inten_out = [[None] * nd for _ in range(ns)] if inten_out is None else [list(r) for r in inten_out]
        inten_scale = [[1.0] * nd for _ in range(ns)] if inten_scale is None else [[float(v) for v in r] for r in inten_scale]
        waves = [[torch.empty(shape, dtype=torch.complex64, device=self.device) if w else None for w in want_wave]
                 for _ in range(ns)]
        for s0 in range(0, ns, _lib.PSX_MAX_SRC):
            sl = range(s0, min(ns, s0 + _lib.PSX_MAX_SRC))
            n = len(sl)
            for s in sl:
                if wave_in is not None and wave_in[s] is not None:
                    _need(wave_in[s], torch.complex64, "wave_in[%d]" % s, shape)
                for d, t in enumerate(inten_out[s]):
                    if t is not None:
                        _need(t, torch.float32, "inten_out[%d][%d]" % (s, d), shape)
            wi = (c_void_p * n)(*[(wave_in[s].data_ptr() if wave_in is not None and wave_in[s] is not None else None) for s in sl])
            wo = (c_void_p * (n * nd))(*[(waves[s][d].data_ptr() if waves[s][d] is not None else None) for s in sl for d in range(nd)])
            io = (c_void_p * (n * nd))(*[(inten_out[s][d].data_ptr() if inten_out[s][d] is not None else None) for s in sl for d in range(nd)])
            check(lib().psx_fresnel_propagate_sources(
                self._h, n, nd, wi, (c_float * n)(*[amp[s] for s in sl]), T,
                (c_double * max(1, n * nm))(*[c for s in sl for c in cph[s]]),
                (c_double * max(1, n * nm))(*[c for s in sl for c in cat[s]]), nm,
                (c_double * (n * nd))(*[float(a[s][d]) for s in sl for d in range(nd)]),
                (c_double * (n * nd))(*[float(gphase[s][d]) for s in sl for d in range(nd)]), c_double(du[0]), c_double(du[1]),
                wo, io, (c_float * (n * nd))(*[inten_scale[s][d] for s in sl for d in range(nd)]), _stream()),
                "psx_fresnel_propagate_sources")
        return waves


# ------------------------------------------------------------------------------------------------- detector
class DetectorPlan:
    """psx_detector_plan: the composed blur/bin/PSF operator of Detector.detection (Detector.py:79-119)."""

    def __init__(self, Nx, Ny, ov, nx, ny, sigma_src, sigma_psf, margin=MARGIN_DETECTOR, device=None):
        self.Nx, self.Ny, self.nx, self.ny = int(Nx), int(Ny), int(nx), int(ny)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._h = c_void_p(None)
        with torch.cuda.device(self.device):
            check(lib().psx_detector_plan_create(self.Nx, self.Ny, int(ov), self.nx, self.ny, int(margin),
                                                 c_double(sigma_src), c_double(sigma_psf), ctypes.byref(self._h)),
                  "psx_detector_plan_create")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().psx_detector_plan_destroy(self._h)
            self._h = c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def detect(self, img, out=None):
        _need(img, torch.float32, "img", (self.Nx, self.Ny))
        if out is None:
            out = torch.empty((self.nx, self.ny), dtype=torch.float32, device=img.device)
        _need(out, torch.float32, "out", (self.nx, self.ny))
        check(lib().psx_detect_f32(self._h, _ptr(img), _ptr(out), _stream()), "psx_detect_f32")
        return out

    def detect_many(self, imgs, outs=None):
        """detect() of several images of this plan's shape -- the images of an energy bin -- PSX_MAX_DETECT per call: fused
        stages take them in one launch each; every image comes out exactly as detect() of it alone would."""
        outs = [None] * len(imgs) if outs is None else list(outs)
        for k, img in enumerate(imgs):
            _need(img, torch.float32, "imgs[%d]" % k, (self.Nx, self.Ny))
            if outs[k] is None:
                outs[k] = torch.empty((self.nx, self.ny), dtype=torch.float32, device=img.device)
            _need(outs[k], torch.float32, "outs[%d]" % k, (self.nx, self.ny))
        for k0 in range(0, len(imgs), _lib.PSX_MAX_DETECT):
            part_in, part_out = imgs[k0:k0 + _lib.PSX_MAX_DETECT], outs[k0:k0 + _lib.PSX_MAX_DETECT]
            pin = (c_void_p * len(part_in))(*[t.data_ptr() for t in part_in])
            pout = (c_void_p * len(part_out))(*[t.data_ptr() for t in part_out])
            check(lib().psx_detect_multi_f32(self._h, pin, pout, len(part_in), _stream()), "psx_detect_multi_f32")
        return outs


def detector_operator_host(N, ov, n, sigma_src, sigma_psf, margin=MARGIN_DETECTOR):
    """One axis of the composed detector operator, built on the host (no GPU): (start[n], weights[n][W])."""
    import numpy as np
    wcap = 4096
    start = (c_int * n)()
    weights = (c_float * (n * wcap))()
    W = c_int(0)
    check(lib().psx_detector_operator_host(int(N), int(ov), int(n), int(margin), c_double(sigma_src), c_double(sigma_psf),
                                           start, weights, wcap, ctypes.byref(W)), "psx_detector_operator_host")
    w = np.frombuffer(weights, dtype=np.float32).reshape(n, wcap)[:, :W.value].copy()
    return np.frombuffer(start, dtype=np.int32).copy(), w


def resize(img, sx, sy):
    """Detector.resize (Detector.py:185-198): identity when sizes match, else s x s block sums, s = Nx//sx."""
    _need(img, torch.float32, "img")
    if img.shape[0] == sx and img.shape[1] == sy:
        return img
    out = torch.empty((sx, sy), dtype=torch.float32, device=img.device)
    check(lib().psx_resize_f32(_ptr(img), img.shape[0], img.shape[1], _ptr(out), int(sx), int(sy), _stream()),
          "psx_resize_f32")
    return out


def poisson(lam, seed):
    """Shot noise (Detector.py:113-115) from a counter-based generator keyed by (seed, pixel)."""
    _need(lam, torch.float32, "lam")
    out = torch.empty_like(lam)
    check(lib().psx_poisson_f32(_ptr(lam), _ptr(out), lam.numel(), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), _stream()),
          "psx_poisson_f32")
    return out


def poisson_key(seed, pointNum=0, ibin=0, kind=0):
    """64-bit generator key of one detector image: a hash of WHAT is drawn -- the user's seed, the membrane position, the
    energy bin and the image kind (0 sample, 1 reference, 2 propagation, 3 white) -- so the noise of an image does not
    depend on call order or on how the positions are sharded over GPUs (splitmix64 finaliser per field)."""
    m = (1 << 64) - 1
    h = 0x9E3779B97F4A7C15
    for v in (int(seed), int(pointNum), int(ibin), int(kind)):
        h = (h ^ (v & m)) & m
        h = (h + 0x9E3779B97F4A7C15) & m
        h = ((h ^ (h >> 30)) * 0xBF58476D1CE4E5B9) & m
        h = ((h ^ (h >> 27)) * 0x94D049BB133111EB) & m
        h = h ^ (h >> 31)
    return h


def poisson_multi(imgs, seeds):
    """In-place shot noise on several equally sized images in ONE launch, image i under key seeds[i]."""
    if not imgs:
        return imgs
    if len(imgs) > _lib.PSX_MAX_POISSON or len(seeds) != len(imgs):
        raise PsxError("poisson_multi: 1..%d images with one key each" % _lib.PSX_MAX_POISSON)
    for i, t in enumerate(imgs):
        _need(t, torch.float32, "imgs[%d]" % i, imgs[0].shape)
    ptr = (c_void_p * len(imgs))(*[t.data_ptr() for t in imgs])
    sd = (ctypes.c_uint64 * len(imgs))(*[int(v) & (2 ** 64 - 1) for v in seeds])
    check(lib().psx_poisson_multi_f32(ptr, sd, len(imgs), imgs[0].numel(), _stream()), "psx_poisson_multi_f32")
    return imgs


def pack_counts(src, dst, index0, exc, exc_count, overflow):
    """dst (int16 storage, read as uint16) = src as 16-bit photon counts; counts >= 65535 leave the escape code 65535 and
    (index0 + p, count) in the exception table exc (int32 [cap, 2]; exc_count int32 [1] counts them); overflow (int32 [1])
    is raised when src is not all integers in [0, 2^24] or the table is full -- dst is then not to be used."""
    _need(src, torch.float32, "src")
    _need(dst, torch.int16, "dst", src.shape)
    _need(exc, torch.int32, "exc")
    _need(exc_count, torch.int32, "exc_count")
    _need(overflow, torch.int32, "overflow")
    check(lib().psx_pack_counts_u16(_ptr(src), _ptr(dst), src.numel(), int(index0), _ptr(exc), _ptr(exc_count),
                                    exc.numel() // 2, _ptr(overflow), _stream()), "psx_pack_counts_u16")
    return dst


def unpack_counts(src, dst, exc=None, exc_count=None):
    """dst (float32) = the 16-bit counts of src (int16 storage, read as uint16), then the exceptions of pack_counts."""
    _need(src, torch.int16, "src")
    _need(dst, torch.float32, "dst", src.shape)
    cap = 0
    if exc is not None:
        _need(exc, torch.int32, "exc")
        _need(exc_count, torch.int32, "exc_count")
        cap = exc.numel() // 2
    check(lib().psx_unpack_counts_u16(_ptr(src), _ptr(dst), src.numel(), _ptr(exc) if cap else None,
                                      _ptr(exc_count) if cap else None, cap, _stream()), "psx_unpack_counts_u16")
    return dst


def status_scan(img):
    _need(img, torch.float32, "img")
    check(lib().psx_status_scan_f32(_ptr(img), img.numel(), _ptr(status_word(img.device)), _stream()),
          "psx_status_scan_f32")
