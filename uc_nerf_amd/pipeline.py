"""The 64 coarse + 128 fine hierarchical renderer composed from the library's entry points
(ray_marcher -> render -> sample_pdf on mid-points with w[1:-1] -> sorted merge -> render).

Library calls per batch (+ the source repack when asked), all on the caller's stream:
  ray_gen_sample (rays + coarse depths) -> render_fused_fwd (coarse; its compositing launch also runs sample_pdf + merge) -> render_fused_fwd (fine).
  Five kernel launches per step on the gather-fused route (rays, MLP, composite + re-sample, MLP, composite); `fold_rays` moves the first into the
  coarse MLP launch (four launches; measured slower, off by default).
"""
import torch

from . import _lib as L
from . import ops


def flat_params_of(state_dict):
    """Concatenates a reference-style UCNeRF state_dict (state_dict order) into the flat vector the packer reads."""
    return torch.cat([v.reshape(-1) for v in state_dict.values()]).float()


class CoarseFineRenderer:
    """scene: dict with K[3,3], c2w[4,4] (target camera), w2cs[V,4,4] + intrinsics[V,3,3] (index 0 = reference
    view, 1.. = source views), near, far, vols (3 x [1,8,D,h,w]), imgs [1,V-1,3,H,W], img_feat [V-1,1,8,H,W],
    confidence [H,W]; tensors already on the device.  flat_params: the MLP's flat parameter vector (device)."""

    def __init__(self, scene, flat_params, n_coarse=64, n_fine=128, white_bkgd=False, pe_layout=0, max_blocks=0,
                 precision="f32", fused_min_rounds=0, sources_bf16=False):
        dev = scene["confidence"].device
        self.scene, self.dev = scene, dev
        self.n_coarse, self.n_fine, self.white_bkgd, self.max_blocks = n_coarse, n_fine, white_bkgd, max_blocks
        self.src = ops.GatherSources(scene["vols"], scene["confidence"], scene["imgs"], scene["img_feat"],
                                     scene["w2cs"][1:], scene["intrinsics"][1:], cl_bf16=sources_bf16)
        self.precision = precision          # "f32": exact fp32 MFMA; "bf16x3": split-bf16 matrix cores (inference, within the
                                            # parity bar); "bf16": plain bf16 operands (inference, ~3e-3 render error);
                                            # "bf16x3_fused": bf16x3 with the feature gather inside the MLP kernel (one launch per pass)
        self.pw = ops.PackedWeights.get(self.src.V, pe_layout, dev, precision)
        self.wstream = self.pw.pack(flat_params)
        w2c_ref = scene["w2cs"][0]
        self.pass_ = ops.RenderPass(self.src, self.pw, self.wstream, scene["c2w"][:3, 3].to(dev), w2c_ref,
                                    scene["intrinsics"][0], w2c_ref, scene["near"], scene["far"], white_bkgd, max_blocks)
        # fused_min_rounds > 0: passes with fewer tiles per wave than that take the two-kernel route (a second stream of the same
        # parameters).  Measured (scripts/ab_rounds.sh, 512 .. 2048 rays x 64 + 128): the fused kernel wins or ties at every size,
        # so the default is 0 -- always fused; the option stays for devices / shapes where a launch-bound shard might prefer otherwise.
        self.pass_small, self.fused_min_samples = None, 0
        if precision == "bf16x3_fused" and fused_min_rounds > 0:
            cus = torch.cuda.get_device_properties(dev).multi_processor_count if dev.type == "cuda" else 256
            self.fused_min_samples = int(fused_min_rounds) * cus * 8 * 32
            self.pw_small = ops.PackedWeights.get(self.src.V, pe_layout, dev, "bf16x3")
            self.wstream_small = self.pw_small.pack(flat_params)
            self.pass_small = ops.RenderPass(self.src, self.pw_small, self.wstream_small, scene["c2w"][:3, 3].to(dev), w2c_ref,
                                             scene["intrinsics"][0], w2c_ref, scene["near"], scene["far"], white_bkgd, max_blocks)
        self.u_det = torch.linspace(0., 1., n_fine, device=dev)
        # camera matrices and depth range travel BY VALUE in the ABI structs: keep host copies so that a render call
        # never reads device memory back (a read-back would drain the stream once per batch)
        self.K_host, self.c2w_host = scene["K"].detach().cpu(), scene["c2w"].detach().cpu()
        self.w2c_dir_host = w2c_ref.detach().cpu()
        self.near_host, self.far_host = float(scene["near"]), float(scene["far"])
        self.sampler = ops.RaySampler(self.K_host, self.c2w_host, n_coarse, self.near_host, self.far_host, self.w2c_dir_host)
        # round 4: the small launches around the coarse pass are folded into it -- ray_gen_sample into the gather-fused kernel's prologue, the coarse
        # compositing + sample_pdf + merge into one launch (a 512-ray shard's step is five launches instead of seven); False restores the old structure
        self.fold_launches = True     # coarse compositing + sample_pdf + merge: one launch (-2.5 us per step at 512 and at 4096 rays, same box)
        # ray_gen_sample inside the gather-fused coarse launch (ABI v4 gen_rays / gen_depths).  Per tile (the RAYGEN instantiation) it is bit-identical
        # and measured SLOWER than the 4.8-us launch it removes -- +1.2 us per step at 512 rays, +8 us at 4096 (profiles/r04_experiments.md: five
        # correctly rounded divisions and a square root per lane and tile, twice, are ~200 vector instructions in a kernel where none is free).  On the
        # tail route of small passes a block owns whole rays and makes them once, in its prologue: free.  None = that case only.
        self.fold_rays = None

    def set_params(self, flat_params):
        self.wstream.copy_(self.pw.pack(flat_params))
        if self.pass_small is not None:
            self.wstream_small.copy_(self.pw_small.pack(flat_params))

    def _pass_for(self, n_samples):
        """The render pass serving a pass of `n_samples` samples (see fused_min_samples)."""
        if self.pass_small is not None and n_samples < self.fused_min_samples:
            if self.pass_.use_cl and not self.pass_small.use_cl:
                self.pass_small.repack_sources(force=False)      # (the channel-last copies belong to the shared sources object)
            return self.pass_small
        return self.pass_

    def render(self, xs, ys, perturb=0.0, noise=None, u=None, events=None, repack=True, reuse_coarse=False):
        """xs, ys: pixel coordinates [n] (device, float32).  events: optional [(start, stop), (start, stop)]
        Event pairs recorded around the coarse and the fine MLP launches.  repack: rebuild the channel-last source
        copies first (needed whenever volumes / images / features changed since the last call).
        reuse_coarse: the fine pass evaluates the network on the n_fine NEW depths only and takes the n_coarse coarse
        depths' outputs from the coarse pass (a sample's output depends on nothing but that sample, so the merged
        rows -- and everything composited from them -- are bit-identical to re-evaluating all n_coarse + n_fine, which is
        what the reference and the default do): one third less network and gather work in the fine pass."""
        sc = self.scene
        if repack:
            self.pass_.repack_sources()
        # rays and their view-direction feature from one launch; both passes take the feature as an input
        n = int(xs.shape[0])
        cpass = self._pass_for(n * self.n_coarse)
        # fold_rays (gather-fused kernel, up to six source views): the coarse launch generates rays, coarse depths and direction features itself
        # (ABI v4 gen_rays / gen_depths) -- no ray_gen_sample launch
        fold_rays = self.fold_rays
        if fold_rays is None:
            fold_rays = (cpass.pw.cfg.precision == 3 and cpass.use_cl and not self.src.cl_bf16 and not self.max_blocks
                         and bool(L.lib().ucnerf_fused_tail_fits_resample(n, self.n_coarse, self.n_fine) if self.fold_launches
                                  else L.lib().ucnerf_fused_tail_fits(n, self.n_coarse)))
        gen = self.sampler if (fold_rays and cpass.pw.cfg.precision == 3 and cpass.use_cl and self.src.V <= 6) else None
        rays_d, angle, z_c = self.sampler.prepare(xs, ys, perturb, noise) if gen is not None else self.sampler(xs, ys, perturb, noise)
        ev = [(a.h, b.h) for a, b in events] if events else (None, None)
        # the coarse pass's compositing launch draws the fine depths as well (composite + sample_pdf + sorted merge: one launch, ABI v4)
        if self.fold_launches:
            coarse = cpass(rays_d, z_c, want=("weights",), events=ev[0], keep=("raw",) if reuse_coarse else (), dir_feat=angle, gen=gen,
                           resample={"u": self.u_det if u is None else u, "want_rank": reuse_coarse})
            hs = coarse
        else:                                                   # (the launch structure of rounds 1-3, kept for A/B and the bit-identity tests)
            coarse = cpass(rays_d, z_c, want=("weights",), events=ev[0], keep=("raw",) if reuse_coarse else (), dir_feat=angle, gen=gen)
            hs = ops.sample_pdf(None, coarse["weights"], self.u_det if u is None else u, z_merge=z_c, want_inds=False,
                                from_coarse=True, want_rank=reuse_coarse)
        if reuse_coarse:
            new = self._pass_for(n * self.n_fine)(rays_d, hs["samples"], want=(), events=ev[1], keep=("raw",), dir_feat=angle)
            raw = ops.merge_rows(new["raw"], coarse["raw"], hs["merge_rank"])      # cat(samples, z_coarse) order
            out = ops.composite_fwd(raw, hs["z_sorted"], 0, self.white_bkgd)
        else:
            out = self._pass_for(n * (self.n_coarse + self.n_fine))(rays_d, hs["z_sorted"], want=("acc", "weights", "var"), events=ev[1], dir_feat=angle)
        out.update(z_coarse=z_c, z_fine=hs["z_sorted"], z_samples=hs["samples"], coarse=coarse, rays_d=rays_d)
        return out

    # ------------------------------------------------------------------------------------------------ HIP graph
    def capture(self, n_rays, perturb=0.0, repack=True, reuse_coarse=False):
        """Captures one render of `n_rays` rays into a HIP graph (the launch-bound regime of small
        per-GPU batches).  Returns a callable g(xs, ys, noise=None) -> the same dict as render(); its tensors are owned
        by the graph and overwritten by the next replay.  Every launch of the step goes to the capturing stream and
        the step allocates only through torch's caching allocator, so the capture is a plain stream capture."""
        dev = self.dev
        xs_s, ys_s = torch.zeros(n_rays, device=dev), torch.zeros(n_rays, device=dev)
        noise_s = torch.rand(n_rays, self.n_coarse, device=dev) if perturb > 0 else None
        kw = dict(perturb=perturb, noise=noise_s, repack=repack, reuse_coarse=reuse_coarse)
        from .train_step import _CAPTURE_STREAMS          # one stream of the graph's own for warm-up and capture, never destroyed (train_step.GraphedStep)
        cap = torch.cuda.Stream(device=dev)
        _CAPTURE_STREAMS.append(cap)
        cap.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(cap):                        # warm-up: one-time attribute calls, workspace allocation
            for _ in range(2):
                self.render(xs_s, ys_s, **kw)
        torch.cuda.current_stream(dev).wait_stream(cap)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=cap):
            out = self.render(xs_s, ys_s, **kw)

        def replay(xs=None, ys=None, noise=None):
            """One hipGraphLaunch.  Arguments given are first copied into the graph's static inputs (one small launch each); a caller
            that wants none of those writes into `replay.inputs` itself and calls replay()."""
            if xs is not None:
                xs_s.copy_(xs, non_blocking=True)
            if ys is not None:
                ys_s.copy_(ys, non_blocking=True)
            if noise_s is not None and noise is not None:
                noise_s.copy_(noise, non_blocking=True)
            graph.replay()
            return out

        replay.graph = graph
        replay.inputs = {"xs": xs_s, "ys": ys_s, "noise": noise_s}
        return replay
