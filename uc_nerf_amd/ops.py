"""Torch-facing wrappers of the HIP entry points: tensors in, tensors out, on the caller's current stream.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); every computation is a kernel of
libucnerf_hip.so.  All functions require float32 tensors on a ROCm device and raise otherwise.
"""
import ctypes as C
import os

import torch

from . import _lib as L


# ------------------------------------------------------------------------------------------------ helpers
def _dev(t):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise RuntimeError("uc_nerf_amd: expected a tensor on a ROCm device, got %s"
                           % (t.device if torch.is_tensor(t) else type(t)))
    return t.device


def _f32(t, name="tensor"):
    _dev(t)
    if t.dtype != torch.float32:
        raise RuntimeError("uc_nerf_amd: %s must be float32, got %s" % (name, t.dtype))
    return t.contiguous()


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# the raw handle of torch's current stream / the current device index: torch.cuda.current_stream() builds a Stream object (~5 us a call);
# the public calls are the fallback where a torch build lacks the two private ones
_cur_dev = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda idx: torch.cuda.current_stream(idx).cuda_stream)


def _stream():
    return _raw_stream(_cur_dev())


def _mat(dst, src, rows, cols):
    """Copies the top-left rows x cols block of a (host or device) matrix into a ctypes float array."""
    m = torch.as_tensor(src, dtype=torch.float32).detach().cpu()
    for r in range(rows):
        for c in range(cols):
            dst[r * cols + c] = float(m[r, c])


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on(device):
    """Context that makes `device` current for the library call inside: nothing at all when it already is (one process per GPU)."""
    idx = device.index
    return _NO_SWITCH if idx is None or idx == _cur_dev() else torch.cuda.device(device)


def _launch(name, params, device):
    with _on(device):                                     # (no switch, no context manager when the device is current: one process per GPU)
        L.call(name, params, _stream())


# ------------------------------------------------------------------------------------------------ a1/a2
def ray_gen(K, c2w, xs=None, ys=None, H=0, W=0, grid_start=0, n=None, opengl=False, want_origin=False, want_pix=False,
            device=None, w2c_dir=None):
    """Pinhole rays.  Either (xs, ys) pixel lists or a row-major grid range [grid_start, grid_start+n) of an
    HxW image.  Returns (rays_d[n,3], rays_o[n,3] or None, pix[2,n] or None); with `w2c_dir` a fourth element, the
    view-direction feature (d/|d|) @ R_dir^T [n,3] (what `dir_feature(rays_d, w2c_dir)` gives, from the same launch)."""
    p = L.RayGenParams()
    if xs is not None:
        xs, ys = _f32(xs, "xs"), _f32(ys, "ys")
        device, n = xs.device, xs.numel()
        p.xs, p.ys = _ptr(xs), _ptr(ys)
    else:
        if device is None:
            raise RuntimeError("uc_nerf_amd.ray_gen: grid mode needs a device")
        n = H * W - grid_start if n is None else n
    p.n, p.H, p.W, p.grid_start, p.opengl = n, H, W, grid_start, int(opengl)
    if opengl:
        p.K[0] = float(K)
    else:
        _mat(p.K, K, 3, 3)
    _mat(p.c2w, c2w, 3, 4)
    rays_d = torch.empty(n, 3, device=device)
    rays_o = torch.empty(n, 3, device=device) if want_origin else None
    pix = torch.empty(2, n, device=device) if want_pix else None
    p.rays_d, p.rays_o, p.pix = _ptr(rays_d), _ptr(rays_o), _ptr(pix)
    angle = None
    if w2c_dir is not None:
        _mat(p.w2c_dir, w2c_dir, 3, 4)
        angle = torch.empty(n, 3, device=device)
        p.angle = _ptr(angle)
    _launch("ucnerf_ray_gen", p, device)
    return (rays_d, rays_o, pix) if w2c_dir is None else (rays_d, rays_o, pix, angle)


class RaySampler:
    """ray_gen(xs, ys) + sample_stratified(n, S, near, far) from ONE launch, camera matrices bound once (they travel by value in
    the parameter structs: filling them per call costs more host time than the launch).  Call -> (rays_d [n,3], angle [n,3] or
    None, z [n,S])."""

    def __init__(self, K, c2w, S, near, far, w2c_dir=None, lindisp=False):
        self.rg, self.ss = L.RayGenParams(), L.SampleStratifiedParams()
        _mat(self.rg.K, K, 3, 3)
        _mat(self.rg.c2w, c2w, 3, 4)
        self.has_dir = w2c_dir is not None
        if self.has_dir:
            _mat(self.rg.w2c_dir, w2c_dir, 3, 4)
        self.ss.S, self.ss.lindisp, self.ss.near, self.ss.far = int(S), int(lindisp), float(near), float(far)

    def prepare(self, xs, ys, perturb=0.0, noise=None):
        """Binds inputs and freshly allocated outputs WITHOUT launching: (rays_d, angle, z) are filled by whoever runs the bound structs --
        `__call__` (ucnerf_ray_gen_sample) or a render pass handed this sampler as `gen=` (ABI v4: the gather-fused MLP launch of the coarse pass
        generates rays and depths in its own prologue)."""
        xs, ys = _f32(xs, "xs"), _f32(ys, "ys")
        n, device = xs.numel(), xs.device
        rg, ss = self.rg, self.ss
        rays_d = torch.empty(n, 3, device=device)
        angle = torch.empty(n, 3, device=device) if self.has_dir else None
        z = torch.empty(n, ss.S, device=device)
        rg.n, rg.xs, rg.ys, rg.rays_d, rg.angle = n, _ptr(xs), _ptr(ys), _ptr(rays_d), _ptr(angle)
        ss.n, ss.perturb, ss.z = n, float(perturb), _ptr(z)
        if perturb > 0:
            noise = _f32(noise if noise is not None else torch.rand(n, ss.S, device=device), "noise")
        ss.noise = _ptr(noise if perturb > 0 else None)
        self._alive = (xs, ys, noise)              # (until the launch that reads them is enqueued)
        return rays_d, angle, z

    def __call__(self, xs, ys, perturb=0.0, noise=None):
        out = self.prepare(xs, ys, perturb, noise)
        with _on(out[0].device):
            L.check(L.lib().ucnerf_ray_gen_sample(C.addressof(self.rg), C.addressof(self.ss), _stream()), "ucnerf_ray_gen_sample")
        return out


def ray_gen_sample(K, c2w, xs, ys, S, near, far, perturb=0.0, noise=None, lindisp=False, w2c_dir=None):
    return RaySampler(K, c2w, S, near, far, w2c_dir, lindisp)(xs, ys, perturb, noise)


def ndc_rays(H, W, focal_x, focal_y, near, rays_o, rays_d, variant):
    rays_o, rays_d = _f32(rays_o.reshape(-1, 3)), _f32(rays_d.reshape(-1, 3))
    p = L.NdcRaysParams()
    p.n, p.H, p.W, p.variant = rays_o.shape[0], int(H), int(W), variant
    p.focal_x, p.focal_y, p.near = float(focal_x), float(focal_y), float(near)
    out_o, out_d = torch.empty_like(rays_o), torch.empty_like(rays_d)
    p.rays_o, p.rays_d, p.out_o, p.out_d = _ptr(rays_o), _ptr(rays_d), _ptr(out_o), _ptr(out_d)
    _launch("ucnerf_ndc_rays", p, rays_o.device)
    return out_o, out_d


def dir_feature(rays_d, w2c_ref=None):
    """Returns (angle[n,3], cos_angle[n]): normalised direction rotated into the reference camera.  A float32 w2c_ref
    already on the rays' device ([3,4] or [4,4], contiguous) is read by the kernel in place -- no host read-back."""
    rays_d = _f32(rays_d, "rays_d")
    p = L.DirFeatureParams()
    p.n, p.has_ref = rays_d.shape[0], int(w2c_ref is not None)
    if (torch.is_tensor(w2c_ref) and w2c_ref.is_cuda and w2c_ref.dtype == torch.float32 and w2c_ref.device == rays_d.device
            and w2c_ref.dim() == 2 and w2c_ref.shape[0] >= 3 and w2c_ref.shape[1] == 4 and w2c_ref.is_contiguous()):
        p.w2c_ref_dev = _ptr(w2c_ref)
    elif w2c_ref is not None:
        _mat(p.w2c_ref, w2c_ref, 3, 4)
    angle = torch.empty_like(rays_d)
    cos = torch.empty(rays_d.shape[0], device=rays_d.device)
    p.rays_d, p.angle, p.cos_angle = _ptr(rays_d), _ptr(angle), _ptr(cos)
    _launch("ucnerf_dir_feature", p, rays_d.device)
    return angle, cos


# ------------------------------------------------------------------------------------------------ a3
def sample_stratified(rays, S, lindisp=False, perturb=0.0, noise=None, want_pts=True, n=None, near=None, far=None,
                      device=None):
    """ray_marcher depths.  Either `rays` [n,8] (per-ray near/far, optional points) or n + scalar near/far."""
    p = L.SampleStratifiedParams()
    if rays is not None:
        rays = _f32(rays, "rays")
        n, device = rays.shape[0], rays.device
    else:
        p.near, p.far, want_pts = float(near), float(far), False
    p.n, p.S, p.lindisp, p.perturb = n, int(S), int(lindisp), float(perturb)
    if perturb > 0:
        if noise is None:
            noise = torch.rand(n, S, device=device)
        noise = _f32(noise, "noise")
    z = torch.empty(n, S, device=device)
    pts = torch.empty(n, S, 3, device=device) if want_pts else None
    p.rays, p.noise, p.z, p.pts = _ptr(rays), _ptr(noise if perturb > 0 else None), _ptr(z), _ptr(pts)
    _launch("ucnerf_sample_stratified", p, device)
    return z, pts


def sample_cascade(near_far, S, t_rand=None, rays_o=None, rays_d=None):
    """near_far [n,6] per-ray cascade ranges -> sorted + stratified depths [n,S] (and points when rays given)."""
    near_far = _f32(near_far, "near_far")
    n = near_far.shape[0]
    p = L.SampleCascadeParams()
    p.n, p.S = n, int(S)
    t_rand = _f32(t_rand) if t_rand is not None else None
    z = torch.empty(n, S, device=near_far.device)
    pts = None
    if rays_d is not None:
        rays_o, rays_d = _f32(rays_o.reshape(-1)[:3]), _f32(rays_d)
        pts = torch.empty(n, S, 3, device=near_far.device)
    p.near_far, p.t_rand, p.rays_o, p.rays_d, p.z, p.pts = (_ptr(near_far), _ptr(t_rand), _ptr(rays_o), _ptr(rays_d),
                                                            _ptr(z), _ptr(pts))
    _launch("ucnerf_sample_cascade", p, near_far.device)
    return z, pts


# ------------------------------------------------------------------------------------------------ a4
def ndc_project(pts, w2c, K, inv_scale, near_far=None, sample_2d=False):
    """get_ndc_coordinate.  pts [N,S,3].  near_far: dict with near_1..far_3 ([N,S,1] tensors or scalars) and
    near/far scalars.  Returns dict(stage1, stage2, stage3, ndc) or a single tensor when sample_2d."""
    pts = _f32(pts, "pts")
    shp = pts.shape
    m = pts.numel() // 3
    p = L.NdcProjectParams()
    p.m, p.has_w2c, p.sample_2d = m, int(w2c is not None), int(sample_2d)
    if w2c is not None:
        _mat(p.w2c, w2c, 3, 4)
    _mat(p.K, K, 3, 3)
    s = torch.as_tensor(inv_scale, dtype=torch.float32).detach().cpu().reshape(-1)
    p.inv_scale[0], p.inv_scale[1] = float(s[0]), float(s[1])
    p.pts = _ptr(pts)
    dev = pts.device
    if sample_2d:
        out = torch.empty(shp, device=dev)
        p.out_ndc = _ptr(out)
        _launch("ucnerf_ndc_project", p, dev)
        return out
    keep, strides = [], set()
    for k in ("near_1", "far_1", "near_2", "far_2", "near_3", "far_3"):
        v = near_far[k]
        if torch.is_tensor(v) and v.numel() == m:
            v = _f32(v.to(dev).reshape(-1))
            strides.add(1)
        else:
            v = torch.as_tensor(v, dtype=torch.float32).reshape(-1)[:1].to(dev).contiguous()
            strides.add(0)
        keep.append(v)
        setattr(p, k, _ptr(v))
    if len(strides) != 1:
        raise RuntimeError("uc_nerf_amd.ndc_project: stage near/far must be all per-sample or all scalar")
    p.nf_stride = strides.pop()
    p.near, p.far = float(near_far["near"]), float(near_far["far"])
    outs = {k: torch.empty(shp, device=dev) for k in ("stage1", "stage2", "stage3", "ndc")}
    p.out_stage1, p.out_stage2, p.out_stage3, p.out_ndc = (_ptr(outs["stage1"]), _ptr(outs["stage2"]),
                                                           _ptr(outs["stage3"]), _ptr(outs["ndc"]))
    _launch("ucnerf_ndc_project", p, dev)
    return outs


def _dev_f32(t, name, numel=None, dev=None):
    """`t` as it is when it is a contiguous float32 tensor on the device (read in place by a kernel); one conversion otherwise."""
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        t = torch.as_tensor(t, dtype=torch.float32, device=dev).contiguous()
        if not t.is_cuda:
            raise RuntimeError("uc_nerf_amd: %s must live on a ROCm device" % name)
    if numel is not None and t.numel() < numel:
        raise RuntimeError("uc_nerf_amd: %s has %d elements, expected at least %d" % (name, t.numel(), numel))
    return t


_BRT_PREPARED = {}       # (ids + data pointers of the camera / hypothesis tensors, image size, S) -> (params struct with the constant fields set, the tensors kept alive)


def build_rays_test(H, W, grid_start, n, S, K, c2w, w2c_ref, K_ref, near_far_ref, depth_values, t_rand=None, want_ranges=False):
    """The evaluation loop's ray builder in ONE launch (utils/utils.py:600-739; ucnerf_build_rays_test): rays through pixels grid_start ..
    grid_start + n - 1 of the HxW grid, cascade ranges from the three depth_values [.., D_k, h_k, w_k], sorted + jittered depths, world points
    and the four normalised copies.  Every matrix is a DEVICE tensor read in place (K [3,3], c2w [4,4] or [3,4], w2c_ref, K_ref, near_far_ref
    [2]): nothing is read back to the host.  Called 80 times per image behind a drained stream, so its host time counts: the argument checks and
    the constant half of the parameter struct are done once per set of input tensors, the five coordinate arrays are one allocation.
    -> dict(rays_o [3], rays_d [n,3], z [n,S], pts / stage1 / stage2 / stage3 / ndc [n,S,3] (+ ranges [n,6]))."""
    dev = c2w.device
    # (keyed by ADDRESS, element type and -- for the hypothesis volumes -- shape: the caller's per-call views of the same memory, e.g.
    #  near_fars[ref_idx], are new objects every time; a key that does not match in full takes the checked route below)
    try:
        key = (int(H), int(W), int(S)) + tuple((t.data_ptr(), t.dtype, t.shape, t.is_contiguous()) for t in (K, c2w, w2c_ref, K_ref, near_far_ref, *depth_values))
    except AttributeError:              # (something that is not a tensor: converted below, not cached)
        key = None
    prep = _BRT_PREPARED.get(key) if key is not None else None
    if prep is None:
        K_, c2w_, w2c_, Kr_ = _dev_f32(K, "K", 9, dev), _dev_f32(c2w, "c2w", 12, dev), _dev_f32(w2c_ref, "w2c_ref", 12, dev), _dev_f32(K_ref, "K_ref", 9, dev)
        nf_ = _dev_f32(near_far_ref, "near_far_ref", 2, dev)
        if c2w_.shape[-1] != 4 or w2c_.shape[-1] != 4:
            raise RuntimeError("uc_nerf_amd.build_rays_test: c2w / w2c_ref must have 4 columns")
        p = L.BuildRaysTestParams()
        p.S, p.H, p.W = int(S), int(H), int(W)
        keep = [K_, c2w_, w2c_, Kr_, nf_]
        for k, dv in enumerate(depth_values):
            dv = _dev_f32(dv, "depth_values")
            keep.append(dv)
            p.dv_d[k], p.dv_h[k], p.dv_w[k] = dv.shape[-3], dv.shape[-2], dv.shape[-1]
            if dv.numel() != dv.shape[-3] * dv.shape[-2] * dv.shape[-1]:
                raise RuntimeError("uc_nerf_amd.build_rays_test: depth_values must be [1,D,h,w] (one batch entry)")
            p.depth_values[k] = dv.data_ptr()
        p.K, p.c2w, p.w2c_ref, p.K_ref, p.near_far_ref = K_.data_ptr(), c2w_.data_ptr(), w2c_.data_ptr(), Kr_.data_ptr(), nf_.data_ptr()
        prep = (p, keep)
        # cached only when every input was usable as it is (the prepared struct then points at the caller's own memory, which the key identifies)
        if key is not None and keep[0] is K and keep[1] is c2w and keep[2] is w2c_ref and keep[3] is K_ref and keep[4] is near_far_ref and all(a is b_ for a, b_ in zip(keep[5:], depth_values)):
            if len(_BRT_PREPARED) >= 8:        # (a handful of scenes at most: the cache must not grow with a caller that builds new tensors per call)
                _BRT_PREPARED.clear()
            _BRT_PREPARED[key] = prep
    p = prep[0]
    n, m = int(n), int(n) * int(S)
    if t_rand is not None:
        t_rand = _f32(t_rand, "t_rand")
        if t_rand.numel() != m:
            raise RuntimeError("uc_nerf_amd.build_rays_test: t_rand must be [n,S]")
    coords = torch.empty(5, n, int(S), 3, device=dev)                  # pts, stage1, stage2, stage3, ndc: one allocation, five views from one call
    pts, s1, s2, s3, ndc = coords.unbind(0)
    z, rays_d, rays_o = torch.empty(n, int(S), device=dev), torch.empty(n, 3, device=dev), torch.empty(3, device=dev)
    ranges = torch.empty(n, 6, device=dev) if want_ranges else None
    p.n, p.grid_start, p.t_rand = n, int(grid_start), _ptr(t_rand)
    p.rays_o, p.rays_d, p.near_far, p.z = rays_o.data_ptr(), rays_d.data_ptr(), _ptr(ranges), z.data_ptr()
    base, step = coords.data_ptr(), 12 * m
    p.pts, p.ndc1, p.ndc2, p.ndc3, p.ndc = base, base + step, base + 2 * step, base + 3 * step, base + 4 * step
    _launch("ucnerf_build_rays_test", p, dev)
    out = {"rays_o": rays_o, "rays_d": rays_d, "z": z, "pts": pts, "stage1": s1, "stage2": s2, "stage3": s3, "ndc": ndc}
    if want_ranges:
        out["ranges"] = ranges
    return out


# ------------------------------------------------------------------------------------------------ a5
def embed(x, n_freqs, layout=0):
    x = _f32(x, "x")
    if x.shape[-1] != 3:
        raise RuntimeError("uc_nerf_amd.embed: last dimension must be 3")
    out = torch.empty(*x.shape[:-1], 3 + 6 * n_freqs, device=x.device)
    p = L.EmbedParams()
    p.m, p.n_freqs, p.layout, p.x, p.out = x.numel() // 3, n_freqs, layout, _ptr(x), _ptr(out)
    _launch("ucnerf_embed", p, x.device)
    return out


# ------------------------------------------------------------------------------------------------ a7
def _strides_match(t, want):
    """Strides of `t` equal `want` on every dimension that has more than one element (torch leaves the others arbitrary)."""
    return all(n == 1 or st == w for n, st, w in zip(t.shape, t.stride(), want))


def _cl_ok(t):
    return torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32


def cl_volume_view(v):
    """`v` ([1,8,D,h,w] or [8,D,h,w]) as the [8,D,h,w] view the kernels can read IN PLACE, when its memory is [D,h,w,8] (what torch calls
    channels_last_3d: a Conv3d running in that memory format writes it) and 16-byte aligned; None otherwise."""
    if not _cl_ok(v) or v.dim() not in (4, 5) or (v.dim() == 5 and v.shape[0] != 1):
        return None
    t = v[0] if v.dim() == 5 else v
    c, d, h, w = t.shape
    if c != 8 or t.data_ptr() % 16 or not _strides_match(t, (1, h * w * 8, w * 8, 8)):
        return None
    return t


def cl_img_feat_view(f):
    """`f` ([V,1,8,H,W] or [V,8,H,W]) as the [V,8,H,W] view readable in place: memory [V,H,W,8] (channels_last of [V,8,H,W]), 16-byte aligned."""
    if not _cl_ok(f) or f.dim() not in (4, 5) or (f.dim() == 5 and f.shape[1] != 1):
        return None
    t = f[:, 0] if f.dim() == 5 else f
    V, c, H, W = t.shape
    if c != 8 or t.data_ptr() % 16 or not _strides_match(t, (H * W * 8, 1, W * 8, 8)):
        return None
    return t


def cl_imgs_view(im):
    """`im` ([1,V,3,H,W] or [V,3,H,W]) as ([V,3,H,W] view, values per pixel) when its memory is [V,H,W,3] (channels_last of [V,3,H,W]) or
    [V,H,W,4] (padded pixels, 16-byte aligned); None otherwise."""
    if not _cl_ok(im) or im.dim() not in (4, 5) or (im.dim() == 5 and im.shape[0] != 1):
        return None
    t = im[0] if im.dim() == 5 else im
    V, c, H, W = t.shape
    if c != 3:
        return None
    for px, align in ((3, 4), (4, 16)):
        if t.data_ptr() % align == 0 and _strides_match(t, (H * W * px, 1, W * px, px)):
            return t, px
    return None


class ChannelLastSources:
    """The gather's sources allocated in the layouts the kernels read, as torch tensors in the reference's SHAPES -- a convenience for a producer
    that can write its outputs there (or for sources that stay fixed over many steps); any tensor with these strides is treated the same way,
    wherever it was allocated (e.g. the output of an nn.Conv3d running in channels_last_3d):

        vols[k]   [1, 8, D, h, w]   memory [D,h,w,8] = torch.channels_last_3d                    <- volume_feature_no_ref of stage k + 1
        imgs      [1, V, 3, H, W]   memory [V,H,W,3] = torch.channels_last of [V,3,H,W]          <- imgs[:, 1:]
        img_feat  [V, 1, 8, H, W]   memory [V,H,W,8] = torch.channels_last of [V,8,H,W]          <- img_feat

    Handed to `rendering()` / `GatherSources` (as they are, or `.detach().requires_grad_()` of them), they are recognised by their strides: the
    pass reads them ZERO-COPY -- no per-step repack (25 us, 150 MB of traffic) -- and returns the source gradients with the same strides (no
    transposing pass either).  Every source is recognised on its own: those that do not qualify take the repack route, the others stay in place."""

    def __init__(self, vol_dhw, V, H, W, device):
        self.vol_dhw, self.V, self.H, self.W = [tuple(int(x) for x in t) for t in vol_dhw], int(V), int(H), int(W)
        self.vols = [torch.zeros(1, 8, d, h, w, device=device).contiguous(memory_format=torch.channels_last_3d) for d, h, w in self.vol_dhw]
        self.imgs = torch.zeros(V, 3, H, W, device=device).contiguous(memory_format=torch.channels_last).unsqueeze(0)
        self.img_feat = torch.zeros(V, 8, H, W, device=device).contiguous(memory_format=torch.channels_last).unsqueeze(1)

    @classmethod
    def from_reference_layout(cls, vols, imgs, img_feat):
        """Allocates channel-last tensors for these sources (reference layouts: volumes [1,8,D,h,w], imgs [1,V,3,H,W], img_feat [V,1,8,H,W]) and
        copies them in -- once; afterwards the new tensors ARE the sources."""
        dhw = [tuple(v.shape[-3:]) for v in vols]
        imgs4 = imgs.reshape(-1, 3, *imgs.shape[-2:])
        c = cls(dhw, imgs4.shape[0], imgs.shape[-2], imgs.shape[-1], imgs.device)
        with torch.no_grad():
            for dst, src in zip(c.vols, vols):
                dst.copy_(src.reshape(dst.shape))
            c.imgs.copy_(imgs.reshape(c.imgs.shape))
            c.img_feat.copy_(img_feat.reshape(c.img_feat.shape))
        return c


class GatherSources:
    """The gather's read-only geometry + source tensors, normalised to the layouts the kernels read.
    Any of `vols`, `conf`, `imgs` may be None: the corresponding units are masked out (their feature columns are
    left unwritten), which is how index_point_feature / build_color_volume run on the same kernel."""

    def __init__(self, vols, conf, imgs, img_feat, w2cs, intrinsics, hw=None, cl_bf16=False):
        """cl_bf16: the channel-last copies the fast gather reads (RenderPass.repack_sources) hold bf16 instead of fp32 -- SURVEY.md 8
        configs[4] "fp32 MLP / bf16 features": half the bytes per gather corner, features within bf16 rounding (2^-9 relative) of the
        fp32 sources'; source gradients are still accumulated in fp32 for the fp32 tensors."""
        self.cl_bf16 = bool(cl_bf16)
        self.mask = 0
        dev = None
        self.vols = [None, None, None]
        # PER SOURCE (ABI v5): a tensor whose memory already is the channel-last layout the fast kernels read -- a channels_last_3d volume, a
        # channels_last feature / image stack, each its own allocation -- is read IN PLACE (no repack, ever) and its gradient comes back with the
        # same strides.  Its reference-layout pointer is withheld (fill()): an entry point that reads channel-major sources fails loudly instead
        # of reading the wrong layout.  inplace[k]: 0..2 volumes, 3 img_feat, 4 imgs.
        self.inplace = [False] * 5
        self.rgb_stride = 4
        if vols is not None:
            if any(v is None for v in vols):
                raise RuntimeError("uc_nerf_amd: three cascade volumes or none")
            for k, v in enumerate(vols):
                t = None if cl_bf16 else cl_volume_view(v)
                self.inplace[k] = t is not None
                self.vols[k] = t if t is not None else _f32(v, "volume").reshape(v.shape[-4:])       # [8,D,h,w]
                if self.vols[k].shape[0] != 8:
                    raise RuntimeError("uc_nerf_amd: cascade volumes must have 8 channels")
            self.mask |= 0b111
            dev = self.vols[0].device
        self.conf = None
        if conf is not None:
            hw = tuple(conf.shape[-2:])
            self.conf = _f32(conf, "confidence").reshape(hw)
            self.mask |= 0b1000
            dev = self.conf.device
        self.imgs = self.img_feat = self.w2cs = self.intrinsics = None
        self.V = 1
        if imgs is not None:
            hw = tuple(imgs.shape[-2:])
            t = None if cl_bf16 else cl_imgs_view(imgs)
            if t is not None:
                self.imgs, self.rgb_stride = t
                self.inplace[4] = True
            else:
                self.imgs = _f32(imgs, "imgs").reshape(-1, 3, *hw)                     # [V,3,H,W]
            self.V = self.imgs.shape[0]
            dev = self.imgs.device
            t = None if (cl_bf16 or img_feat is None) else cl_img_feat_view(img_feat)
            if t is not None and t.shape[0] == self.V and tuple(t.shape[2:]) == hw:
                self.img_feat = t
                self.inplace[3] = True
            else:
                self.img_feat = (_f32(img_feat, "img_feat").reshape(self.V, 8, *hw) if img_feat is not None
                                 else torch.zeros(self.V, 8, *hw, device=dev))
            self.w2cs = torch.as_tensor(w2cs, dtype=torch.float32)[:, :3, :4].reshape(-1, 12).to(dev).contiguous()
            self.intrinsics = torch.as_tensor(intrinsics, dtype=torch.float32).reshape(-1, 9).to(dev).contiguous()
            if self.w2cs.shape[0] != self.V or self.intrinsics.shape[0] != self.V:
                raise RuntimeError("uc_nerf_amd: %d source images but %d poses / %d intrinsics"
                                   % (self.V, self.w2cs.shape[0], self.intrinsics.shape[0]))
            self.mask |= ((1 << self.V) - 1) << 4
        if hw is None or self.mask == 0:
            raise RuntimeError("uc_nerf_amd: the gather needs at least one source")
        self.H, self.W = int(hw[0]), int(hw[1])
        self.F = 24 + 12 * self.V + 1
        self.device = dev
        self.full = self.mask == (0b1111 | (((1 << self.V) - 1) << 4))
        self.zero_copy = all(self.inplace)          # every source is read in place: nothing to repack, now or later
        # ucnerf_cl_sources of the in-place sources alone, and of all five once RenderPass.repack_sources has completed them (shared by every
        # RenderPass bound to these sources); _cl: the buffer the repacked ones live in
        self.cl_inplace = L.ClSources()
        for k in range(3):
            self.cl_inplace.vol[k] = _ptr(self.vols[k]) if self.inplace[k] else None
        self.cl_inplace.img_feat = _ptr(self.img_feat) if self.inplace[3] else None
        self.cl_inplace.imgs = _ptr(self.imgs) if self.inplace[4] else None
        self.cl_inplace.rgb_stride, self.cl_inplace.bf16 = self.rgb_stride, int(self.cl_bf16)
        self.cl_all = self.cl_inplace if self.zero_copy else None
        self._cl = None

    def fill(self, p):
        p.V, p.H, p.W = self.V, self.H, self.W
        p.unit_mask = 0 if self.full else self.mask
        for k, v in enumerate(self.vols):
            if v is not None:
                p.vol_d[k], p.vol_h[k], p.vol_w[k] = v.shape[1], v.shape[2], v.shape[3]
            else:
                p.vol_d[k] = p.vol_h[k] = p.vol_w[k] = 1
            p.vol[k] = None if self.inplace[k] else _ptr(v)
        p.conf = _ptr(self.conf)
        p.imgs, p.img_feat = (None if self.inplace[4] else _ptr(self.imgs)), (None if self.inplace[3] else _ptr(self.img_feat))
        p.w2cs, p.intrinsics = _ptr(self.w2cs), _ptr(self.intrinsics)


def _opt(t):
    return _f32(t) if t is not None else None


def feat_gather_fwd(src, pts, ndc1, ndc2, ndc3, tiled=False, u_out=None):
    """u_out: optional [m] float32 tensor receiving the per-sample uncertainty 1 - sampled confidence."""
    pts, ndc1, ndc2, ndc3 = _opt(pts), _opt(ndc1), _opt(ndc2), _opt(ndc3)
    lead = next(t for t in (pts, ndc1, ndc3) if t is not None)
    m = lead.numel() // 3
    p = L.FeatGatherParams()
    src.fill(p)
    p.m, p.out_tiled = m, int(tiled)
    n_out = ((m + 31) // 32) * 32 * src.F if tiled else m * src.F
    feats = torch.empty(n_out, device=lead.device) if src.full else torch.zeros(n_out, device=lead.device)
    p.pts, p.ndc1, p.ndc2, p.ndc3, p.feats = _ptr(pts), _ptr(ndc1), _ptr(ndc2), _ptr(ndc3), _ptr(feats)
    if u_out is not None:
        if u_out.numel() != m or u_out.dtype != torch.float32 or not u_out.is_contiguous() or src.conf is None:
            raise RuntimeError("uc_nerf_amd.feat_gather_fwd: u_out must be a contiguous float32 [m] tensor (and the confidence map given)")
        p.u_out = _ptr(u_out)
    _launch("ucnerf_feat_gather_fwd", p, lead.device)
    return feats if tiled else feats.view(*lead.shape[:-1], src.F)


def feat_gather_bwd(src, pts, ndc1, ndc2, ndc3, g_feats, need=(True, True, True, True, True)):
    """Returns grads (g_vol1, g_vol2, g_vol3, g_conf, g_img_feat); entries not needed are None."""
    pts, ndc1, ndc2, ndc3, g_feats = _opt(pts), _opt(ndc1), _opt(ndc2), _opt(ndc3), _f32(g_feats)
    lead = next(t for t in (pts, ndc1, ndc3) if t is not None)
    if any(src.inplace):
        raise RuntimeError("uc_nerf_amd.feat_gather_bwd: channel-last sources are served by the render passes (RenderPass.backward)")
    bp = L.FeatGatherBwdParams()
    src.fill(bp.fwd)
    bp.fwd.m = lead.numel() // 3
    bp.fwd.pts, bp.fwd.ndc1, bp.fwd.ndc2, bp.fwd.ndc3 = _ptr(pts), _ptr(ndc1), _ptr(ndc2), _ptr(ndc3)
    bp.g_feats = _ptr(g_feats)
    gv = [torch.zeros_like(v) if (need[k] and v is not None) else None for k, v in enumerate(src.vols)]
    gc = torch.zeros_like(src.conf) if (need[3] and src.conf is not None) else None
    gi = torch.zeros_like(src.img_feat) if (need[4] and src.imgs is not None) else None
    for k in range(3):
        bp.g_vol[k] = _ptr(gv[k])
    bp.g_conf, bp.g_img_feat = _ptr(gc), _ptr(gi)
    bp.scratch = _ptr(_gather_scratch(src, bp.fwd, lead.device))
    _launch("ucnerf_feat_gather_bwd", bp, lead.device)
    return gv[0], gv[1], gv[2], gc, gi


def _gather_scratch(src, fwd_params, device):
    """Channel-last accumulation buffer of the gather backward, cached on the sources object."""
    n = L.lib().ucnerf_feat_gather_bwd_scratch_floats(C.addressof(fwd_params))
    buf = getattr(src, "_bwd_scratch", None)
    if buf is None or buf.numel() < n or buf.device != device:
        buf = torch.empty(max(int(n), 4), device=device)
        src._bwd_scratch = buf
    return buf


class _FeatGather(torch.autograd.Function):
    """Differentiable w.r.t. the three volumes, confidence and img_feat (never positions: SURVEY 3.2).
    Sources given as None are skipped (see GatherSources)."""

    @staticmethod
    def forward(ctx, vol1, vol2, vol3, conf, img_feat, imgs, w2cs, intrinsics, pts, ndc1, ndc2, ndc3):
        vols = None if vol1 is None else [vol1, vol2, vol3]
        src = GatherSources(vols, conf, imgs, img_feat, w2cs, intrinsics)
        ctx.src = src
        ctx.shapes = tuple(None if t is None else t.shape for t in (vol1, vol2, vol3, conf, img_feat))
        ctx.coords = (pts, ndc1, ndc2, ndc3)
        return feat_gather_fwd(src, pts, ndc1, ndc2, ndc3)

    @staticmethod
    def backward(ctx, g):
        pts, ndc1, ndc2, ndc3 = ctx.coords
        need = ctx.needs_input_grad[:5]
        grads = feat_gather_bwd(ctx.src, pts, ndc1, ndc2, ndc3, g.reshape(-1, ctx.src.F), need)
        out = [x.reshape(s) if (x is not None and s is not None) else None for x, s in zip(grads, ctx.shapes)]
        return tuple(out) + (None,) * 7


def feat_gather(vols, conf, img_feat, imgs, w2cs, intrinsics, pts, ndc1, ndc2, ndc3):
    return _FeatGather.apply(vols[0], vols[1], vols[2], conf, img_feat, imgs, w2cs, intrinsics, pts, ndc1, ndc2, ndc3)


# ------------------------------------------------------------------------------------------------ a6
BACKWARD_MODES = {"chain": 0, "layerwise": 1}
_backward_mode = BACKWARD_MODES[os.environ.get("UCNERF_BWD_MODE", "chain")]


def set_backward_mode(mode):
    """MLP backward: "chain" (default) = the register-resident gradient chain (one kernel walks the network backwards per 32-sample tile,
    split-bf16 data gradients, include/ucnerf_hip.h: bwd_mode 0), "layerwise" = the exact-fp32 layer-by-layer GEMMs (bwd_mode 1)."""
    global _backward_mode
    _backward_mode = BACKWARD_MODES[mode]


SPLIT_OPERANDS = {"bf16": 0, "fp16": 1}
_split_operand = SPLIT_OPERANDS[os.environ.get("UCNERF_SPLIT_OPERAND", "bf16")]


def set_split_operand(kind):
    """The 16-bit terms of the split precisions ("bf16x3", "bf16x3_fused", "bf16") from now on (ucnerf_mlp_config.operand, ABI v6): "bf16" (default:
    8 significant bits per term, float32's range) or "fp16" (11 bits per term at the same matrix-core rate -- the three-product split then holds ~22
    bits and the renders sit at float32 level, no measurable kernel time (+0.1 %) -- but fp16's range: an activation beyond 131 008 is clamped, a term below 6e-5 is held
    to 3e-8 absolute).  Takes effect
    for every weight stream packed afterwards (PackedWeights.get keys on it); inference only."""
    global _split_operand
    if kind not in SPLIT_OPERANDS:
        raise ValueError("uc_nerf_amd: split operand must be 'bf16' or 'fp16', got %r" % (kind,))
    _split_operand = SPLIT_OPERANDS[kind]


def split_operand():
    return "fp16" if _split_operand else "bf16"


class PackedWeights:
    """Pack index (host-built by the library, cached on device) + packing of a flat parameter vector."""
    _cache = {}

    PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 2, "bf16x3_fused": 3}     # 3: bf16x3 with the gather inside the MLP kernel (render passes only)

    def __init__(self, n_src, pe_layout, device, precision="f32", operand=0):
        self.cfg = L.MlpConfig(n_src, pe_layout, self.PRECISIONS[precision], 0 if precision == "f32" else int(operand))
        self.precision, self.operand = precision, self.cfg.operand
        lib = L.lib()
        self.n_params = lib.ucnerf_mlp_param_count(C.addressof(self.cfg))
        self.n_stream = lib.ucnerf_mlp_stream_count(C.addressof(self.cfg))
        n_idx = lib.ucnerf_mlp_index_count(C.addressof(self.cfg))
        if self.n_params < 0 or self.n_stream < 0 or n_idx < 0:
            raise RuntimeError("uc_nerf_amd: unsupported MLP config n_src=%d precision=%s: %s"
                               % (n_src, precision, lib.ucnerf_last_error().decode()))
        idx = torch.empty(n_idx, dtype=torch.int32)
        L.check(lib.ucnerf_mlp_pack_index(C.addressof(self.cfg), C.c_void_p(idx.data_ptr())), "ucnerf_mlp_pack_index")
        self.idx_host = idx
        self.idx = idx.to(device)
        self.device = device

    @classmethod
    def get(cls, n_src, pe_layout, device, precision="f32", operand=None):
        """operand: None = the module's current setting (set_split_operand); "bf16" / "fp16" / 0 / 1 to pin it."""
        op = _split_operand if operand is None else SPLIT_OPERANDS.get(operand, operand)
        op = 0 if precision == "f32" else int(op)
        key = (n_src, pe_layout, str(device), precision, op)
        if key not in cls._cache:
            cls._cache[key] = cls(n_src, pe_layout, device, precision, op)
        return cls._cache[key]

    def pack(self, flat, out=None):
        """The packed stream of a flat parameter vector; into `out` (a stream buffer of this packer) when given."""
        flat = _f32(flat, "flat parameters")
        if flat.numel() != self.n_params:
            raise RuntimeError("uc_nerf_amd: flat parameter vector has %d floats, expected %d"
                               % (flat.numel(), self.n_params))
        if out is None:
            out = torch.empty(self.n_stream, device=flat.device)
        elif out.numel() != self.n_stream or out.dtype != torch.float32 or not out.is_contiguous() or out.device != flat.device:
            raise RuntimeError("uc_nerf_amd: stream buffer must be %d contiguous float32 on the parameters' device" % self.n_stream)
        with _on(flat.device):
            L.check(L.lib().ucnerf_mlp_pack(C.addressof(self.cfg), _ptr(flat), _ptr(self.idx), _ptr(out), _stream()), "ucnerf_mlp_pack")
        return out

    def pack_table(self, table, out):
        """Packs the stream straight from separate parameter tensors (TensorTable) into `out`, in place: one launch, no concatenation."""
        if table.total != self.n_params:
            raise RuntimeError("uc_nerf_amd: the parameter tensors hold %d floats, expected %d" % (table.total, self.n_params))
        if out.numel() != self.n_stream or out.dtype != torch.float32 or not out.is_contiguous():
            raise RuntimeError("uc_nerf_amd: stream buffer must be %d contiguous float32" % self.n_stream)
        with _on(out.device):
            L.check(L.lib().ucnerf_mlp_pack_tensors(C.addressof(self.cfg), table.n, table.ptrs, table.numel, _ptr(self.idx), _ptr(out), _stream()),
                    "ucnerf_mlp_pack_tensors")
        return out

    def unpack_grad(self, g_stream):
        g_flat = torch.zeros(self.n_params, device=g_stream.device)
        with _on(g_stream.device):
            L.check(L.lib().ucnerf_mlp_unpack_grad(_ptr(g_stream), _ptr(self.idx), _ptr(g_flat), self.n_stream, _stream()),
                    "ucnerf_mlp_unpack_grad")
        return g_flat


class TensorTable:
    """Host-side table of device pointers / element counts of a network's parameter tensors (state_dict order) for
    PackedWeights.pack_table; `key` = the data pointers it was built from."""

    def __init__(self, tensors):
        tensors = list(tensors)
        for t in tensors:
            if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("uc_nerf_amd.TensorTable: parameters must be contiguous float32 tensors on a ROCm device")
        self.n = len(tensors)
        self.key = tuple(t.data_ptr() for t in tensors)
        self.ptrs = (C.c_void_p * self.n)(*self.key)
        self.numel = (C.c_int64 * self.n)(*[t.numel() for t in tensors])
        self.total = sum(t.numel() for t in tensors)


def mlp_fwd(pw, wstream, pts, dirs, feats, S, feats_tiled=False, max_blocks=0):
    """pts [m,3] (any leading shape), dirs [m/S,3] or [m,3], feats [m,F] or tiled buffer -> raw [m,4]."""
    pts, dirs, feats = _f32(pts, "pts"), _f32(dirs, "dirs"), _f32(feats, "feats")
    m = pts.numel() // 3
    p = L.MlpParams()
    p.cfg = pw.cfg
    p.m, p.S = m, int(S)
    n_dirs = dirs.numel() // 3
    if n_dirs == m:
        p.dirs_per_sample = 1
    elif S > 0 and n_dirs * S == m:
        p.dirs_per_sample = 0
    else:
        raise RuntimeError("uc_nerf_amd.mlp_fwd: %d directions for %d samples (S=%d)" % (n_dirs, m, S))
    F = 24 + 12 * pw.cfg.n_src + 1
    need = ((m + 31) // 32) * 32 * F if feats_tiled else m * F
    if feats.numel() != need:
        raise RuntimeError("uc_nerf_amd.mlp_fwd: feats has %d floats, expected %d" % (feats.numel(), need))
    p.feats_tiled, p.max_blocks = int(feats_tiled), int(max_blocks)
    raw = torch.empty(m, 4, device=pts.device)
    p.pts, p.dirs, p.feats, p.wstream, p.raw = _ptr(pts), _ptr(dirs), _ptr(feats), _ptr(wstream), _ptr(raw)
    _launch("ucnerf_mlp_fwd", p, pts.device)
    return raw


KEPT_SETS = ("bd", "h0", "h1", "h2", "h3", "h4", "h5", "bc", "ft", "vc")       # order of the activation sets at the head of the backward workspace


def decode_p24(buf, m, cols=128):
    """fp32 [m, cols] view-copy of a set kept in the 24-bit format of csrc/p24.h (rows of 3 * cols bytes: the top three bytes of every fp32)."""
    b = buf.view(torch.uint8)[: m * cols * 3].view(m, cols, 3).to(torch.int32)
    bits = (b[..., 0] << 8) | (b[..., 1] << 16) | (b[..., 2] << 24)
    return bits.view(torch.float32)


def mlp_fwd_train(pw, wstream, pts, dirs, feats, S, bwd_mode=None):
    """ucnerf_mlp_fwd_train: the forward that keeps its activation sets for ucnerf_mlp_bwd (tests / diagnostics).  Returns (raw, sets): the ten
    [m,128] sets decoded to fp32, whatever format the backward mode keeps them in (0 / "chain": 24-bit floats, 1 / "layerwise": fp32)."""
    pts, dirs, feats = _f32(pts), _f32(dirs), _f32(feats)
    mode = _backward_mode if bwd_mode is None else BACKWARD_MODES[bwd_mode] if isinstance(bwd_mode, str) else int(bwd_mode)
    m = pts.numel() // 3
    p = L.MlpParams()
    p.cfg = pw.cfg
    p.m, p.S = m, int(S)
    n_dirs = dirs.numel() // 3
    p.dirs_per_sample = int(n_dirs == m)
    if not p.dirs_per_sample and n_dirs * int(S) != m:
        raise RuntimeError("uc_nerf_amd.mlp_fwd_train: %d directions for %d samples (S=%d)" % (n_dirs, m, S))
    raw = torch.empty(m, 4, device=pts.device)
    ws = torch.zeros(L.lib().ucnerf_mlp_bwd_workspace_floats(C.addressof(pw.cfg), m), device=pts.device)
    p.pts, p.dirs, p.feats, p.wstream, p.raw = _ptr(pts), _ptr(dirs), _ptr(feats), _ptr(wstream), _ptr(raw)
    with _on(pts.device):
        L.check(L.lib().ucnerf_mlp_fwd_train(C.addressof(p), _ptr(ws), mode, _stream()), "ucnerf_mlp_fwd_train")
    per = (m * 128 + 3) // 4 * 4
    sets = {}
    for i, name in enumerate(KEPT_SETS):
        chunk = ws[i * per:(i + 1) * per]
        sets[name] = decode_p24(chunk, m) if mode == 0 else chunk[: m * 128].view(m, 128).clone()
    return raw, sets


def mlp_bwd(pw, wstream, flat, pts, dirs, feats, S, g_raw):
    """Gradients of sum(raw * g_raw) w.r.t. feats [m,F] and the flat parameter vector."""
    pts, dirs, feats, g_raw, flat = _f32(pts), _f32(dirs), _f32(feats), _f32(g_raw), _f32(flat)
    m = pts.numel() // 3
    F = 24 + 12 * pw.cfg.n_src + 1
    bp = L.MlpBwdParams()
    p = bp.fwd
    p.cfg = pw.cfg
    p.m, p.S = m, int(S)
    n_dirs = dirs.numel() // 3
    if n_dirs == m:
        p.dirs_per_sample = 1
    elif S > 0 and n_dirs * S == m:
        p.dirs_per_sample = 0
    else:
        raise RuntimeError("uc_nerf_amd.mlp_bwd: %d directions for %d samples (S=%d)" % (n_dirs, m, S))
    if feats.numel() != m * F or g_raw.numel() != m * 4:
        raise RuntimeError("uc_nerf_amd.mlp_bwd: feats must be [m,%d] row-major and g_raw [m,4]" % F)
    need = L.lib().ucnerf_mlp_bwd_workspace_floats(C.addressof(pw.cfg), m)
    ws = torch.empty(need, device=pts.device)
    g_feats = torch.empty(m, F, device=pts.device)
    g_flat = torch.zeros(pw.n_params, device=pts.device)
    p.pts, p.dirs, p.feats, p.wstream, p.raw = _ptr(pts), _ptr(dirs), _ptr(feats), _ptr(wstream), _ptr(g_raw)
    bp.g_raw, bp.flat_params, bp.g_feats, bp.g_flat, bp.workspace = _ptr(g_raw), _ptr(flat), _ptr(g_feats), _ptr(g_flat), _ptr(ws)
    bp.bwd_mode = _backward_mode
    _launch("ucnerf_mlp_bwd", bp, pts.device)
    return g_feats, g_flat


class _MLP(torch.autograd.Function):
    """raw = MLP(PE(pts), feats, PE(dirs)); differentiable w.r.t. the flat parameters and feats."""

    @staticmethod
    def forward(ctx, flat, feats, pts, dirs, pw, S, ws=None):
        ws = pw.pack(flat) if ws is None else ws        # (a stream already packed from these very parameters may be handed in)
        ctx.pw, ctx.S = pw, S
        ctx.save_for_backward(flat, feats, pts, dirs, ws)
        return mlp_fwd(pw, ws, pts, dirs, feats.reshape(-1, feats.shape[-1]), S)

    @staticmethod
    def backward(ctx, g_raw):
        flat, feats, pts, dirs, ws = ctx.saved_tensors
        g_feats, g_flat = mlp_bwd(ctx.pw, ws, flat, pts, dirs, feats.reshape(-1, feats.shape[-1]), ctx.S, g_raw)
        return g_flat, g_feats.view(feats.shape), None, None, None, None, None


def mlp(flat, feats, pts, dirs, pw, S, wstream=None):
    return _MLP.apply(flat, feats, pts, dirs, pw, S, wstream)


def _encoded_params(p, pw, x):
    """Points pts / feats / dirs of an MlpParams into the columns of x [m, 63 + F + 27]."""
    F = 24 + 12 * pw.cfg.n_src + 1
    X = 63 + F + 27
    if x.shape[-1] != X:
        raise RuntimeError("uc_nerf_amd: encoded MLP input must have %d columns (63 + %d + 27), got %d" % (X, F, x.shape[-1]))
    p.cfg = pw.cfg
    p.m, p.S, p.dirs_per_sample, p.encoded = x.shape[0], 1, 1, 1
    p.pts_stride = p.dirs_stride = p.feat_stride = X
    base = x.data_ptr()
    p.pts, p.feats, p.dirs = base, base + 4 * 63, base + 4 * (63 + F)
    return F, X


def mlp_fwd_encoded(pw, wstream, x):
    x = _f32(x, "x")
    p = L.MlpParams()
    _encoded_params(p, pw, x)
    raw = torch.empty(x.shape[0], 4, device=x.device)
    p.wstream, p.raw = _ptr(wstream), _ptr(raw)
    _launch("ucnerf_mlp_fwd", p, x.device)
    return raw


class _MLPEncoded(torch.autograd.Function):
    """UCNeRF.forward(x) on the reference's pre-encoded 187-wide rows.  Gradients: parameters and the feature
    columns of x; the encoded pts/dir columns get zeros (positions are not differentiable on this path)."""

    @staticmethod
    def forward(ctx, flat, x, pw, ws=None):
        ws = pw.pack(flat) if ws is None else ws
        x = _f32(x, "x")
        if x.requires_grad or flat.requires_grad:
            # the weight-gradient kernel reads its operands in 16-byte pieces: the last piece of the last row (direction columns 24 .. 27 of 27)
            # ends four bytes past the matrix -- keep the copy the backward reads inside a buffer that has them
            buf = torch.empty(x.numel() + 4, device=x.device)
            xp = buf[:x.numel()].view(x.shape)
            xp.copy_(x)
            x = xp
        ctx.pw = pw
        ctx.save_for_backward(flat, x, ws)
        return mlp_fwd_encoded(pw, ws, x)

    @staticmethod
    def backward(ctx, g_raw):
        flat, x, ws = ctx.saved_tensors
        pw = ctx.pw
        g_raw = _f32(g_raw)
        bp = L.MlpBwdParams()
        F, X = _encoded_params(bp.fwd, pw, x)
        m = x.shape[0]
        wsz = L.lib().ucnerf_mlp_bwd_workspace_floats(C.addressof(pw.cfg), m)
        work = torch.empty(wsz, device=x.device)
        g_x = torch.zeros(m, X, device=x.device)
        g_flat = torch.zeros(pw.n_params, device=x.device)
        bp.fwd.wstream, bp.fwd.raw = _ptr(ws), _ptr(g_raw)
        bp.g_raw, bp.flat_params, bp.g_flat, bp.workspace = _ptr(g_raw), _ptr(flat), _ptr(g_flat), _ptr(work)
        bp.g_feats, bp.g_feat_stride = g_x.data_ptr() + 4 * 63, X
        bp.bwd_mode = _backward_mode
        _launch("ucnerf_mlp_bwd", bp, x.device)
        return g_flat, g_x, None, None


def mlp_encoded(flat, x, pw, wstream=None):
    return _MLPEncoded.apply(flat, x, pw, wstream)


def untile_feats(feats, m, F):
    """Row-major [m, F] view-copy of features kept in the MLP's tile layout [ceil(m/32)][F][32] (what a training forward keeps)."""
    return feats.view(-1, F, 32).permute(0, 2, 1).reshape(-1, F)[:m]


# ------------------------------------------------------------------------------------------------ a9
def composite_fwd(raw, z, variant=0, white_bkgd=False, rays_d=None, noise=None, want_var=True, u=None):
    """u: optional [n,S] per-sample uncertainty -> out["wu"] [n] = sum_i w_i u_i (composited uncertainty)."""
    raw, z = _f32(raw, "raw"), _f32(z, "z")
    n, S = z.shape
    dev = z.device
    p = L.CompositeParams()
    p.n, p.S, p.variant, p.white_bkgd = n, S, variant, int(white_bkgd)
    out = dict(rgb=torch.empty(n, 3, device=dev), depth=torch.empty(n, device=dev), acc=torch.empty(n, device=dev),
               disp=torch.empty(n, device=dev), weights=torch.empty(n, S, device=dev))
    if variant == 0 and want_var and S >= 2:
        out["var"] = torch.empty(n, device=dev)
    rays_d = _f32(rays_d) if rays_d is not None else None
    noise = _f32(noise) if noise is not None else None
    p.raw, p.z, p.rays_d, p.noise = _ptr(raw), _ptr(z), _ptr(rays_d), _ptr(noise)
    p.rgb_map, p.depth_map, p.acc_map, p.disp_map = _ptr(out["rgb"]), _ptr(out["depth"]), _ptr(out["acc"]), _ptr(out["disp"])
    p.weights, p.var = _ptr(out["weights"]), _ptr(out.get("var"))
    if u is not None:
        u = _f32(u, "u")
        if tuple(u.shape) != (n, S):
            raise RuntimeError("uc_nerf_amd.composite_fwd: u must be [n,S]")
        out["wu"] = torch.empty(n, device=dev)
        p.u, p.wu = _ptr(u), _ptr(out["wu"])
    _launch("ucnerf_composite_fwd", p, dev)
    return out


def composite_bwd(raw, z, g_rgb=None, g_depth=None, g_acc=None, g_weights=None, white_bkgd=False):
    raw, z = _f32(raw), _f32(z)
    n, S = z.shape
    bp = L.CompositeBwdParams()
    bp.fwd.n, bp.fwd.S, bp.fwd.variant, bp.fwd.white_bkgd = n, S, 0, int(white_bkgd)
    bp.fwd.raw, bp.fwd.z = _ptr(raw), _ptr(z)
    gs = [_f32(g) if g is not None else None for g in (g_rgb, g_depth, g_acc, g_weights)]
    bp.g_rgb, bp.g_depth, bp.g_acc, bp.g_weights = (_ptr(g) for g in gs)
    g_raw = torch.empty_like(raw)
    bp.g_raw = _ptr(g_raw)
    _launch("ucnerf_composite_bwd", bp, z.device)
    return g_raw


class _Composite(torch.autograd.Function):
    """raw2outputs (live variant): differentiable w.r.t. raw through rgb_map, depth_map, acc_map and weights."""

    @staticmethod
    def forward(ctx, raw, z, white_bkgd):
        out = composite_fwd(raw, z, 0, white_bkgd)
        ctx.save_for_backward(raw, z)
        ctx.white_bkgd = white_bkgd
        var = out.get("var", torch.zeros_like(out["acc"]))
        ctx.mark_non_differentiable(out["disp"], var)
        return out["rgb"], out["depth"], out["acc"], out["weights"], out["disp"], var

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_acc, g_weights, _g_disp, _g_var):
        raw, z = ctx.saved_tensors
        g_raw = composite_bwd(raw, z, g_rgb, g_depth, g_acc, g_weights, ctx.white_bkgd)
        return g_raw, None, None


def composite(raw, z, white_bkgd=False):
    return _Composite.apply(raw, z, bool(white_bkgd))


# ------------------------------------------------------------------------------------------------ a8
def sample_pdf(bins, weights, u, z_merge=None, want_inds=True, want_cdf=False, from_coarse=False, want_rank=False):
    """Returns dict(samples[n,M], inds[n,M] int64, cdf[n,L] (opt), z_sorted[n,M+n_merge] (when z_merge given)).
    from_coarse: bins=None, weights = coarse weights [n,S], z_merge = coarse depths [n,S] (mid-point bins, w[1:-1]).
    want_rank (with z_merge): merge_rank[n,M+n_merge] int32, the position in z_sorted of each element of
    cat(samples, z_merge) -- feed it to merge_rows."""
    weights, u = _f32(weights, "weights"), _f32(u, "u")
    if from_coarse:
        n, Lb = weights.shape[0], weights.shape[1] - 1
        if z_merge is None or tuple(z_merge.shape) != tuple(weights.shape):
            raise RuntimeError("uc_nerf_amd.sample_pdf: from_coarse needs z_merge with the weights' shape")
    else:
        bins = _f32(bins, "bins")
        n, Lb = bins.shape
        if weights.shape != (n, Lb - 1):
            raise RuntimeError("uc_nerf_amd.sample_pdf: weights must be [n, n_bins-1]")
    M = u.shape[-1]
    dev = weights.device
    p = L.SamplePdfParams()
    p.n, p.n_bins, p.n_samples, p.from_coarse = n, Lb, M, int(from_coarse)
    if u.numel() == n * M:
        p.u_stride = M
    elif u.numel() == M:
        p.u_stride = 0          # one shared row of draws (e.g. linspace)
    else:
        raise RuntimeError("uc_nerf_amd.sample_pdf: u must be [n, M] or [M]")
    out = {"samples": torch.empty(n, M, device=dev)}
    if want_inds:
        out["inds"] = torch.empty(n, M, dtype=torch.int64, device=dev)
    if want_cdf:
        out["cdf"] = torch.empty(n, Lb, device=dev)
    if z_merge is not None:
        z_merge = _f32(z_merge, "z_merge")
        p.n_merge = z_merge.shape[1]
        out["z_sorted"] = torch.empty(n, M + p.n_merge, device=dev)
        if want_rank:
            out["merge_rank"] = torch.empty(n, M + p.n_merge, dtype=torch.int32, device=dev)
            p.merge_rank = _ptr(out["merge_rank"])
    p.bins, p.weights, p.u, p.z_merge = _ptr(None if from_coarse else bins), _ptr(weights), _ptr(u), _ptr(z_merge)
    p.samples, p.inds, p.cdf, p.z_sorted = _ptr(out["samples"]), _ptr(out.get("inds")), _ptr(out.get("cdf")), _ptr(out.get("z_sorted"))
    _launch("ucnerf_sample_pdf", p, dev)
    return out


def merge_rows(a, b, rank):
    """out[r, rank[r, i]] = cat(a[r], b[r])[i] for rows a [n,na,w], b [n,nb,w] and rank [n,na+nb] (int32)."""
    a, b = _f32(a, "a"), _f32(b, "b")
    n, na, w = a.shape
    nb = b.shape[1]
    if rank.dtype != torch.int32 or tuple(rank.shape) != (n, na + nb) or b.shape[0] != n or b.shape[2] != w:
        raise RuntimeError("uc_nerf_amd.merge_rows: shape / dtype mismatch")
    rank = rank.contiguous()
    out = torch.empty(n, na + nb, w, device=a.device)
    p = L.MergeRowsParams()
    p.n, p.na, p.nb, p.width = n, na, nb, w
    p.a, p.b, p.rank, p.out = _ptr(a), _ptr(b), _ptr(rank), _ptr(out)
    _launch("ucnerf_merge_rows", p, a.device)
    return out


# ------------------------------------------------------------------------------------------------ f2
def _cost_volume_params(feats, proj, depth_values, pad):
    V, Cc, H, W = feats.shape
    D, Hp, Wp = depth_values.shape
    if tuple(proj.shape) != (V, 3, 4) or Hp != H + 2 * pad or Wp != W + 2 * pad:
        raise RuntimeError("uc_nerf_amd.cost_volume: shape mismatch")
    p = L.CostVolumeParams()
    p.V, p.C, p.H, p.W, p.D, p.pad = V, Cc, H, W, D, int(pad)
    p.feats, p.proj, p.depth_values = _ptr(feats), _ptr(proj), _ptr(depth_values)
    return p


def _cost_volume_fwd(feats, proj, depth_values, pad, want_count):
    p = _cost_volume_params(feats, proj, depth_values, pad)
    D, Hp, Wp = depth_values.shape
    var = torch.empty(feats.shape[1], D, Hp, Wp, device=feats.device)
    cnt = torch.empty(D, Hp, Wp, device=feats.device) if want_count else None
    p.variance, p.count = _ptr(var), _ptr(cnt)
    _launch("ucnerf_cost_volume", p, feats.device)
    return var, cnt


class _CostVolumeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, proj, depth_values, pad):
        ctx.save_for_backward(feats, proj, depth_values)
        ctx.pad = pad
        return _cost_volume_fwd(feats, proj, depth_values, pad, False)[0]

    @staticmethod
    def backward(ctx, g_var):
        feats, proj, depth_values = ctx.saved_tensors
        bp = L.CostVolumeBwdParams()
        bp.fwd = _cost_volume_params(feats, proj, depth_values, ctx.pad)
        g_var = _f32(g_var, "g_variance")
        g_feats = torch.zeros_like(feats)
        bp.g_variance, bp.g_feats = _ptr(g_var), _ptr(g_feats)
        _launch("ucnerf_cost_volume_bwd", bp, feats.device)
        return g_feats, None, None, None


def cost_volume(feats, proj, depth_values, pad=0, want_count=False):
    """Variance cost volume of one cascade stage (network/mvs_models.py:609-626 with utils/utils.py:1105-1172 inside).
    feats [V,C,H,W] source-view feature maps, proj [V,3,4] = (src_proj @ ref_proj_inv)[:3], depth_values
    [D,H+2pad,W+2pad].  Returns variance [C,D,H+2pad,W+2pad] (and count [D,Hp,Wp] if asked).  Differentiable with
    respect to `feats` (as in the reference: nearest-neighbour lookup, the grid and the mask count carry no gradient)."""
    feats, proj, depth_values = _f32(feats, "feats"), _f32(proj, "proj"), _f32(depth_values, "depth_values")
    if want_count:
        return _cost_volume_fwd(feats.detach(), proj, depth_values, pad, True)
    return _CostVolumeFn.apply(feats, proj.detach(), depth_values.detach(), int(pad))


def _depth_regress_params(prob_pre, prob_init, depth_values, pad):
    D, Hp, Wp = prob_pre.shape
    if tuple(depth_values.shape) != (D, Hp, Wp) or (prob_init is not None and tuple(prob_init.shape) != (D, Hp, Wp)):
        raise RuntimeError("uc_nerf_amd.depth_regress: shape mismatch")
    p = L.DepthRegressParams()
    p.D, p.Hp, p.Wp, p.pad = D, Hp, Wp, int(pad)
    p.prob_pre, p.prob_init, p.depth_values = _ptr(prob_pre), _ptr(prob_init), _ptr(depth_values)
    return p


class _DepthRegressFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prob_pre, depth_values, prob_init, pad):
        D, Hp, Wp = prob_pre.shape
        dev = prob_pre.device
        p = _depth_regress_params(prob_pre, prob_init, depth_values, pad)
        prob = torch.empty(D, Hp, Wp, device=dev)
        depth = torch.empty(Hp - 2 * pad, Wp - 2 * pad, device=dev)
        conf = torch.empty_like(depth)
        p.prob_volume, p.depth, p.confidence = _ptr(prob), _ptr(depth), _ptr(conf)
        _launch("ucnerf_depth_regress", p, dev)
        ctx.save_for_backward(prob, depth_values)
        ctx.pad, ctx.has_init = pad, prob_init is not None
        ctx.mark_non_differentiable(prob)                 # the reference returns prob_volume.detach() (mvs_models.py:646)
        return prob, depth, conf

    @staticmethod
    def backward(ctx, _g_prob, g_depth, g_conf):
        prob, depth_values = ctx.saved_tensors
        bp = L.DepthRegressBwdParams()
        bp.fwd = _depth_regress_params(prob, None, depth_values, ctx.pad)
        bp.fwd.prob_volume = _ptr(prob)
        g_depth = _f32(g_depth, "g_depth") if g_depth is not None else None
        g_conf = _f32(g_conf, "g_confidence") if g_conf is not None else None
        g_x = torch.empty_like(prob)
        bp.g_depth, bp.g_confidence, bp.g_prob_pre = _ptr(g_depth), _ptr(g_conf), _ptr(g_x)
        _launch("ucnerf_depth_regress_bwd", bp, prob.device)
        return g_x, None, (g_x if ctx.has_init else None), None


def depth_regress(prob_pre, depth_values, prob_init=None, pad=0):
    """softmax over depth, expected depth and 4-tap photometric confidence (network/mvs_models.py:629-646).
    prob_pre, depth_values (and prob_init) [D,Hp,Wp] -> (prob_volume [D,Hp,Wp] (detached, as in the reference), depth
    [H,W], confidence [H,W]); depth and confidence are differentiable with respect to the logits."""
    prob_pre, depth_values = _f32(prob_pre, "prob_pre"), _f32(depth_values, "depth_values")
    prob_init = _f32(prob_init, "prob_init") if prob_init is not None else None
    return _DepthRegressFn.apply(prob_pre, depth_values.detach(), prob_init, int(pad))


# ------------------------------------------------------------------------------------------------ a10
class RenderPass:
    """Pre-bound arguments of ucnerf_render_fused_fwd for one scene; call it with (rays_d, z).

    rays_o / w2c_ref / K_ref / w2c_dir / near / far describe how the pass derives sample coordinates and the view-direction
    feature itself; a caller that hands both over (`coords=`, `dir_feat=`: the rendering() drop-in) may leave them None."""

    def __init__(self, src, pw, wstream, rays_o=None, w2c_ref=None, K_ref=None, w2c_dir=None, near=0.0, far=1.0, white_bkgd=False,
                 max_blocks=0):
        # pw / wstream may be packed for either precision ("f32" or "bf16x3"); the backward needs "f32"
        eye34, eye33 = torch.eye(3, 4), torch.eye(3)
        self.rays_o = (_f32(rays_o.reshape(-1)[:3].clone(), "rays_o") if rays_o is not None else torch.zeros(3, device=src.device))
        self.p = p = L.RenderParams()
        p.max_blocks = int(max_blocks)
        _mat(p.w2c_ref, w2c_ref if w2c_ref is not None else eye34, 3, 4)
        _mat(p.K_ref, K_ref if K_ref is not None else eye33, 3, 3)
        _mat(p.w2c_dir, w2c_dir if w2c_dir is not None else eye34, 3, 4)
        p.near, p.far = float(near), float(far)
        p.rays_o = _ptr(self.rays_o)
        self.set_sources(src)
        self.set_weights(pw, wstream)
        self.set_white_bkgd(white_bkgd)
        self._ws = None

    def set_sources(self, src):
        """Binds (other) gather sources; the fast gather is used once repack_sources() has run for them."""
        self.src = src
        src.fill(self.p)
        self.use_cl = False
        self.p.cl = src.cl_inplace                   # (a copy: the in-place sources, if any; the others' entries stay NULL until repack_sources)
        if src.zero_copy:                            # every source IS channel-last: nothing to repack, now or later
            self.use_cl = True

    def set_weights(self, pw, wstream):
        self.pw, self.wstream = pw, wstream
        self.p.cfg = pw.cfg
        self.p.wstream = _ptr(wstream)

    def set_white_bkgd(self, white_bkgd):
        self.p.white_bkgd = int(bool(white_bkgd))

    def repack_sources(self, force=True):
        """(Re)builds the channel-last copies of the sources that are not handed over in place; call whenever those changed.  The copies
        belong to the sources object: with force=False existing ones (made through any RenderPass bound to them) are reused."""
        src = self.src
        if src.zero_copy:                            # (never written: the arrays are the caller's)
            self.p.cl = src.cl_inplace
            self.use_cl = True
            return
        self.p.cl = src.cl_inplace
        n = L.lib().ucnerf_gather_repack_floats(C.addressof(self.p))
        fresh = src._cl is None or src._cl.numel() != n or src.cl_all is None
        if fresh:
            src._cl = torch.empty(n, device=src.device)
            src.cl_all = L.ClSources()
        if fresh or force:
            with _on(src.device):
                L.check(L.lib().ucnerf_gather_repack(C.addressof(self.p), _ptr(src._cl), C.addressof(src.cl_all), _stream()), "ucnerf_gather_repack")
        self.p.cl = src.cl_all
        self.use_cl = True

    @staticmethod
    def _coords(p, coords, m):
        """Coordinates handed over by the caller (dict pts, stage1, stage2, stage3, ndc; each [n,S,3]) instead of derived
        from (rays_d, z) -- what rendering() of the reference receives.  Returns the tensors to keep alive."""
        if coords is None:
            p.pts_in = p.ndc1_in = p.ndc2_in = p.ndc3_in = p.ndc_in = None
            return ()
        keep = [_f32(coords[k], k) for k in ("pts", "stage1", "stage2", "stage3", "ndc")]
        for t in keep:
            if t.numel() != 3 * m:
                raise RuntimeError("uc_nerf_amd.RenderPass: given coordinates must be [n,S,3]")
        p.pts_in, p.ndc1_in, p.ndc2_in, p.ndc3_in, p.ndc_in = (_ptr(t) for t in keep)
        return keep

    def __call__(self, rays_d, z, near_far=None, want=("acc", "weights", "var"), keep=(), events=None, dir_feat=None, coords=None, resample=None, gen=None,
                 w2c_dir_dev=None):
        """want may also name "u" (per-sample uncertainty u = 1 - sampled confidence [n,S], network/models.py:149) and
        "wu" (its composite sum_i w_i u_i [n]) -- the opt-in uncertainty outputs of SURVEY.md 8(a).
        resample: dict(u=draws [n,M] or [M], want_rank=False) -- the pass's compositing launch also draws the NEXT pass's depths from this pass's
        weights (data/ray_utils.py:216-219: mid-point bins, w[1:-1], sorted merge with z): adds out["samples"] [n,M], out["z_sorted"] [n,S+M]
        (and out["merge_rank"]), the results of sample_pdf(None, out["weights"], u, z_merge=z, from_coarse=True) bit for bit, with no launch of its own.
        gen: a RaySampler whose prepare() returned this call's (rays_d, dir_feat, z): the pass generates them itself (gather-fused kernel only) -- the
        values of RaySampler.__call__, bit for bit, with no launch of its own.
        w2c_dir_dev (with dir_feat=None): the rotation of the view-direction feature as a float32 DEVICE tensor ([3,4] or [4,4], contiguous), read by
        the kernels in place of the by-value w2c_dir -- rendering() holds pose_ref['w2cs'][0] on the device; on the tail route the features are
        then made inside the pass's one launch."""
        rays_d, z = _f32(rays_d, "rays_d"), _f32(z, "z")
        dir_feat = _f32(dir_feat, "dir_feat") if dir_feat is not None else None
        n, S = z.shape
        dev = z.device
        p = self.p
        p.n, p.S = n, S
        need = L.lib().ucnerf_render_workspace_floats(n, S, self.src.V)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=dev)
        near_far = _f32(near_far) if near_far is not None else None
        out = {"rgb": torch.empty(n, 3, device=dev), "depth": torch.empty(n, device=dev)}
        if "acc" in want:
            out["acc"] = torch.empty(n, device=dev)
        if "weights" in want:
            out["weights"] = torch.empty(n, S, device=dev)
        if "var" in want and S >= 2:
            out["var"] = torch.empty(n, device=dev)
        if "raw" in keep:
            out["raw"] = torch.empty(n, S, 4, device=dev)
        # training forward (raw + features kept, exact f32 or split-bf16): the features stay in the MLP's tile layout -- the forward reads
        # them as coalesced 128-byte rows (row-major: 4-byte pieces of 388-byte rows, 2x the launch time) and the backward takes them as they are
        training = "raw" in keep and "feats" in keep and self.pw.cfg.precision in (0, 1)
        if "feats" in keep:
            tiled = training and _backward_mode == 0          # (the layer-by-layer backward reads row-major features)
            out["feats"] = torch.empty((n * S + 31) // 32 * 32 * self.src.F, device=dev) if tiled else torch.empty(n * S, self.src.F, device=dev)
            out["feats_tiled"] = tiled
        p.feats_tiled = int(bool(out.get("feats_tiled", False)))
        if "u" in want or "wu" in want:
            out["u"] = torch.empty(n, S, device=dev)
        if "wu" in want:
            out["wu"] = torch.empty(n, device=dev)
        p.u_sampled, p.wu_map = _ptr(out.get("u")), _ptr(out.get("wu"))
        _alive = self._coords(p, coords, n * S)      # noqa: F841 (kept until the launch is enqueued)
        p.rays_d, p.z, p.near_far, p.workspace = _ptr(rays_d), _ptr(z), _ptr(near_far), _ptr(self._ws)
        p.rgb_map, p.depth_map, p.acc_map = _ptr(out["rgb"]), _ptr(out["depth"]), _ptr(out.get("acc"))
        p.weights, p.var, p.raw, p.feats = _ptr(out.get("weights")), _ptr(out.get("var")), _ptr(out.get("raw")), _ptr(out.get("feats"))
        p.ev_mlp_start, p.ev_mlp_stop = events if events is not None else (None, None)
        p.train_workspace = None
        p.dir_feat = _ptr(dir_feat)
        p.w2c_dir_dev = None
        if w2c_dir_dev is not None and dir_feat is None:
            if not (w2c_dir_dev.is_cuda and w2c_dir_dev.dtype == torch.float32 and w2c_dir_dev.dim() == 2 and w2c_dir_dev.shape[0] >= 3
                    and w2c_dir_dev.shape[1] == 4 and w2c_dir_dev.is_contiguous()):
                raise RuntimeError("uc_nerf_amd.RenderPass: w2c_dir_dev must be a contiguous float32 [3,4] or [4,4] tensor on the device")
            p.w2c_dir_dev = _ptr(w2c_dir_dev)
        p.gen_rays = p.gen_depths = None
        if gen is not None:
            if self.pw.cfg.precision != 3 or coords is not None or near_far is not None or dir_feat is None:
                raise RuntimeError("uc_nerf_amd.RenderPass: gen= needs the gather-fused kernel (precision 'bf16x3_fused'), derived coordinates and the sampler's dir_feat")
            p.gen_rays, p.gen_depths = C.addressof(gen.rg), C.addressof(gen.ss)
        p.resample = None
        if resample is not None:
            u = _f32(resample["u"], "u")
            M = u.shape[-1]
            sp = self._resample_p = getattr(self, "_resample_p", None) or L.SamplePdfParams()
            sp.n, sp.n_bins, sp.n_samples, sp.n_merge, sp.from_coarse = n, S - 1, M, S, 1
            if u.numel() == n * M:
                sp.u_stride = M
            elif u.numel() == M:
                sp.u_stride = 0
            else:
                raise RuntimeError("uc_nerf_amd.RenderPass: resample draws must be [n, M] or [M]")
            out["samples"] = torch.empty(n, M, device=dev)
            out["z_sorted"] = torch.empty(n, S + M, device=dev)
            sp.merge_rank = None
            if resample.get("want_rank"):
                out["merge_rank"] = torch.empty(n, S + M, dtype=torch.int32, device=dev)
                sp.merge_rank = _ptr(out["merge_rank"])
            sp.u, sp.samples, sp.z_sorted = _ptr(u), _ptr(out["samples"]), _ptr(out["z_sorted"])
            p.resample = C.addressof(sp)
        self._saved_for = None
        if training:
            # training forward (exact f32 or split-bf16): keep the MLP activations in the backward's workspace so that
            # backward() need not repeat the network forward
            need_b = L.lib().ucnerf_render_bwd_workspace_floats(n, S, self.src.V)
            if getattr(self, "_bwd_ws", None) is None or self._bwd_ws.numel() < need_b:
                self._bwd_ws = torch.empty(need_b, device=dev)
            p.train_workspace = _ptr(self._bwd_ws)
            p.train_bwd_mode = _backward_mode          # (fixes the format the activations are kept in: 24-bit for the chain, fp32 layer by layer)
            self._saved_for = (n, S, out["raw"].data_ptr(), _backward_mode)
        _launch("ucnerf_render_fused_fwd", p, dev)
        return out

    def saved_matches(self, n, S, raw):
        """True when the backward workspace still holds the activations the training forward kept for THIS (n, S, raw) -- no later forward
        overwrote them -- in the format of the current backward mode."""
        return getattr(self, "_saved_for", None) == (n, S, raw.data_ptr(), _backward_mode)

    def backward(self, rays_d, z, kept, g_rgb, g_depth, flat, near_far=None, need=(True, True, True, True, True), coords=None,
                 dir_feat=None, f32_weights=None, flat_room=0):
        """Backward of the last-style forward call: `kept` = its outputs with keep=("raw", "feats").
        Returns (g_flat, g_vol1, g_vol2, g_vol3, g_conf, g_img_feat).  flat_room > n_params: g_flat is returned with that many floats (the
        rest zero) -- room a gradient bucket uses for its scalars and flags without another buffer (flat.FlatStore.grad_room).  A pass bound to bf16x3 weights runs its backward from
        the activations the training forward kept; should another forward have overwritten them, the network forward is
        repeated in f32 and needs `f32_weights` = (PackedWeights, stream) packed from the same parameters."""
        rays_d, z, flat, g_rgb = _f32(rays_d), _f32(z), _f32(flat), _f32(g_rgb)
        n, S = z.shape
        dev = z.device
        bp = L.RenderBwdParams()
        C.memmove(C.addressof(bp.fwd), C.addressof(self.p), C.sizeof(L.RenderParams))
        p = bp.fwd
        p.n, p.S = n, S
        near_far = _f32(near_far) if near_far is not None else None
        p.rays_d, p.z, p.near_far = _ptr(rays_d), _ptr(z), _ptr(near_far)
        p.raw, p.feats = _ptr(kept["raw"]), _ptr(kept["feats"])
        p.feats_tiled = int(bool(kept.get("feats_tiled", False)))
        if p.feats_tiled and _backward_mode != 0:
            raise RuntimeError("uc_nerf_amd.RenderPass.backward: features kept in the tile layout need the gradient chain (set_backward_mode('chain'))")
        p.u_sampled = p.wu_map = None
        _alive = self._coords(p, coords, n * S)      # noqa: F841
        dir_feat = _f32(dir_feat, "dir_feat") if dir_feat is not None else None
        p.dir_feat = _ptr(dir_feat)
        p.ev_mlp_start = p.ev_mlp_stop = None
        saved = self.saved_matches(n, S, kept["raw"])
        ws = self._bwd_ws if saved else torch.empty(L.lib().ucnerf_render_bwd_workspace_floats(n, S, self.src.V), device=dev)
        bp.saved_valid = int(saved)
        bp.bwd_mode = _backward_mode
        if not saved and self.pw.cfg.precision != 0:
            if f32_weights is None:
                raise RuntimeError("uc_nerf_amd.RenderPass.backward: the activations of this bf16x3 training forward were overwritten "
                                   "by a later forward; pass f32_weights=(PackedWeights, stream) to recompute them")
            p.cfg, p.wstream = f32_weights[0].cfg, _ptr(f32_weights[1])
        self._saved_for = None
        # ONE zero fill for all accumulated outputs (six separate torch.zeros were six 5-us launches per step): views of a flat buffer,
        # every segment padded to 16 bytes
        # a source read in place gets its gradient in the SAME layout (channel-last, accumulated straight into it by the gather backward: no scratch,
        # no transposing pass); the pool segment is then viewed with the source's strides -- what autograd's layout contract asks for
        src = self.src
        inp = src.inplace
        shapes = [(max(int(flat_room), self.pw.n_params),)] + \
                 [tuple(v.shape) if need[k] else None for k, v in enumerate(src.vols)] + \
                 [tuple(src.conf.shape) if need[3] else None, tuple(src.img_feat.shape) if need[4] else None]
        sizes = [0 if sh is None else (int(torch.Size(sh).numel()) + 3) // 4 * 4 for sh in shapes]
        pool = torch.zeros(sum(sizes), device=dev)
        segs, off = [], 0
        for sh, sz in zip(shapes, sizes):
            segs.append(None if sh is None else pool[off:off + int(torch.Size(sh).numel())])
            off += sz
        g_flat = segs[0]
        gv = []
        for k in range(3):
            if segs[1 + k] is None:
                gv.append(None)
            elif inp[k]:
                c, d, h, w = src.vols[k].shape
                gv.append(segs[1 + k].view(d, h, w, c).permute(3, 0, 1, 2))
            else:
                gv.append(segs[1 + k].view(shapes[1 + k]))
        gc = None if segs[4] is None else segs[4].view(shapes[4])
        if segs[5] is None:
            gi = None
        elif inp[3]:
            gi = segs[5].view(src.V, src.H, src.W, 8).permute(0, 3, 1, 2)
        else:
            gi = segs[5].view(shapes[5])
        g_depth = _f32(g_depth) if g_depth is not None else None
        bp.g_rgb, bp.g_depth, bp.flat_params, bp.g_flat, bp.workspace = _ptr(g_rgb), _ptr(g_depth), _ptr(flat), _ptr(g_flat), _ptr(ws)
        for k in range(3):
            bp.g_vol[k] = None if inp[k] else _ptr(gv[k])
            bp.g_cl.vol[k] = _ptr(segs[1 + k]) if inp[k] else None
        bp.g_conf = _ptr(gc)
        bp.g_img_feat = None if inp[3] else _ptr(gi)
        bp.g_cl.img_feat = _ptr(segs[5]) if inp[3] else None
        via_scratch = any(gv[k] is not None and not inp[k] for k in range(3)) or (gi is not None and not inp[3])
        bp.gather_scratch = None
        if via_scratch:
            gfwd = L.FeatGatherParams()
            src.fill(gfwd)
            bp.gather_scratch = _ptr(_gather_scratch(src, gfwd, dev))
        _launch("ucnerf_render_fused_bwd", bp, dev)
        return g_flat, gv[0], gv[1], gv[2], gc, gi


# ------------------------------------------------------------------------------------------------ timing
class Event:
    """HIP timing event created by the library (usable with any stream the kernels run on)."""

    def __init__(self):
        self.h = L.lib().ucnerf_event_create()
        if not self.h:
            raise RuntimeError("uc_nerf_amd: event_create failed")

    def record(self):
        L.check(L.lib().ucnerf_event_record(self.h, _stream()), "ucnerf_event_record")

    def elapsed_ms(self, stop):
        ms = C.c_float(0)
        L.check(L.lib().ucnerf_event_elapsed_ms(self.h, stop.h, C.addressof(ms)), "ucnerf_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            L.lib().ucnerf_event_destroy(self.h)
        except Exception:
            pass
