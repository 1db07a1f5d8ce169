"""Drop-in for the hot-path half of the reference's network/models.py:13-283 -- `Embedder`, `get_embedder`,
`weights_init`, `BaseAdapt_Renderer`, `UCNeRF`, `create_ucnerf` -- with the same names, argument orders,
parameter names/shapes (so reference checkpoints load) and return contracts.  All arithmetic runs in
libucnerf_hip.so; tensors must live on a ROCm device.

Not mirrored (out of scope): the legacy MVSNet classes below line 289 and CascadeMVSNet itself --
`create_ucnerf` takes the consistency learner from `args.network_mvs` (any nn.Module) instead of downloading one.
"""
import torch
import torch.nn as nn

from .. import ops
from ..flat import FlatStore

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def weights_init(m):
    """network/models.py:13-17."""
    if isinstance(m, nn.Linear):
        nn.init.kaiming_normal_(m.weight.data)
        if m.bias is not None:
            nn.init.zeros_(m.bias.data)


class Embedder:
    """network/models.py:20-54: [x | sin(2^k x), k < L | cos(2^k x), k < L] (frequency-major "live" layout)."""
    layout = 0

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        self.create_embedding_fn()

    def create_embedding_fn(self):
        kw = self.kwargs
        if not (kw.get("include_input", True) and kw.get("log_sampling", True) and kw.get("input_dims", 3) == 3
                and kw.get("max_freq_log2") == kw.get("num_freqs") - 1):
            raise NotImplementedError("uc_nerf_amd Embedder: only include_input, log-sampled 2^k bands on 3-vectors "
                                      "(the configuration get_embedder builds) are implemented")
        self.n_freqs = int(kw["num_freqs"])
        self.out_dim = 3 + 6 * self.n_freqs
        self.freq_bands = 2. ** torch.arange(self.n_freqs, dtype=torch.float32).reshape(1, -1, 1)

    def embed(self, inputs):
        return ops.embed(inputs, self.n_freqs, self.layout)

    def __call__(self, inputs):
        return self.embed(inputs)


def get_embedder(multires, i=0, input_dims=3):
    """network/models.py:56-71.  The returned callable carries .n_freqs/.layout so run_network_mvs can fuse it."""
    if i == -1:
        return nn.Identity(), 3
    eo = Embedder(include_input=True, input_dims=input_dims, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return eo, eo.out_dim


class BaseAdapt_Renderer(nn.Module):
    """network/models.py:74-184.  Same submodule / parameter names and registration order as the reference, so
    state_dicts are interchangeable; forward() is one fused HIP kernel (plus its backward)."""

    def __init__(self, D=8, W=256, input_ch=3, input_ch_views=3, output_ch=4, input_ch_feat=8, skips=[4],
                 use_viewdirs=False, fine=False, view_num=4):
        super().__init__()
        self.D, self.W, self.input_ch, self.input_ch_views, self.skips = D, W, input_ch, input_ch_views, skips
        self.view_num = view_num - 1
        self.use_viewdirs = use_viewdirs
        self.in_ch_pts, self.in_ch_views, self.in_ch_feat = input_ch, input_ch_views, input_ch_feat
        self.pts_linears = nn.ModuleList(
            [nn.Linear(input_ch, W)] + [nn.Linear(W + input_ch, W) if i in skips else nn.Linear(W, W) for i in range(D - 1)])
        self.pts_bias_depth_fine = nn.Linear(24 + 4 * self.view_num, W)
        self.pts_bias_confidence = nn.Linear(8 * self.view_num, W)
        self.pts_bias_confidence_1 = nn.Linear(1, 1)
        self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        self.view_confi_linears = nn.ModuleList([nn.Linear(input_ch_views + W, W // 2)])
        if not use_viewdirs:
            raise NotImplementedError("uc_nerf_amd: only the use_viewdirs=True head layout UCNeRF builds is implemented")
        self.feature_linear = nn.Linear(W, W)
        self.feature_linear_1 = nn.Linear(W, W)
        self.confi_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W // 2, 1)
        self.alpha_linear_1 = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        self.confi_rgb_linear = nn.Linear(W, 3)
        # the reference's init policy (models.py:107-118): these get kaiming-normal / zero bias, the remaining two
        # (pts_bias_confidence, alpha_linear_1) keep nn.Linear's default
        for mod in (self.pts_bias_depth_fine, self.pts_linears, self.views_linears, self.view_confi_linears,
                    self.confi_linear, self.pts_bias_confidence_1, self.feature_linear, self.feature_linear_1,
                    self.alpha_linear, self.rgb_linear, self.confi_rgb_linear):
            mod.apply(weights_init)
        self._check_supported()
        FlatStore.of(self)                     # the parameters become views of ONE flat buffer (uc_nerf_amd/flat.py); names and shapes unchanged

    def _check_supported(self):
        ok = (self.D == 6 and self.W == 128 and list(self.skips) == [4] and self.input_ch == 63 and self.input_ch_views == 27
              and 1 <= self.view_num <= 8 and self.in_ch_feat == 24 + 12 * self.view_num + 1)
        if not ok:
            raise NotImplementedError(
                "uc_nerf_amd: the HIP MLP is built for the reference's shipped architecture (netdepth 6, netwidth 128, "
                "skip [4], multires 10/4 -> 63/27 inputs, feat_dim = 24 + 12*(view_num-1) + 1, view_num 2..9); got "
                "D=%d W=%d skips=%s in=(%d,%d,%d) view_num=%d" % (self.D, self.W, self.skips, self.input_ch,
                                                                 self.in_ch_feat, self.input_ch_views, self.view_num + 1))

    def _apply(self, fn, *args, **kwargs):
        """`.to()` / `.cuda()` / `.float()` re-point every parameter at a converted copy: gather them into one flat buffer again."""
        out = super()._apply(fn, *args, **kwargs)
        if "_ucnerf_flat_store" in self.__dict__:
            self.__dict__["_ucnerf_flat_store"].sync()
        return out

    # ---- helpers shared with the fused render path
    def flat_parameters(self):
        """All parameters concatenated in state_dict order (autograd-aware): what the weight packer reads."""
        return torch.cat([p.reshape(-1) for p in self.parameters()])

    def packer(self, pe_layout=0):
        dev = next(self.parameters()).device
        return ops.PackedWeights.get(self.view_num, pe_layout, dev)

    def forward_alpha(self, x):
        # the reference's forward_alpha reads self.pts_bias, which does not exist (AttributeError there too)
        raise AttributeError("'BaseAdapt_Renderer' object has no attribute 'pts_bias'")

    def _flat_and_stream(self, pe_layout):
        """(flat parameter vector for autograd, packer, packed stream).  The stream -- and, when no gradient is wanted, the
        flat vector too -- comes from the per-network cache, rebuilt only when a parameter changed (dropin.FusedSession)."""
        from .. import dropin
        if torch.is_grad_enabled():
            if any(p.requires_grad for p in self.parameters()):
                flat = self.flat_parameters()                 # autograd-aware; the stream is packed from this very vector
                pw = self.packer(pe_layout)
                return flat, pw, pw.pack(flat.detach().float())
            # frozen network, gradients wanted for the inputs: the backward takes its data gradients from `flat`, so it must be the REAL
            # parameters of this forward -- a snapshot of the flat buffer, packed now
            return dropin.session_of(self).packed("f32", pe_layout, fresh=True)
        return dropin.session_of(self).packed("f32", pe_layout)

    def forward(self, x, pe_layout=0):
        """x [..., 63 + F + 27] = [encoded pts | features | encoded dirs] -> [..., 4] (rgb, sigma)."""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        flat, pw, ws = self._flat_and_stream(pe_layout)
        out = ops.mlp_encoded(flat, x2, pw, ws)
        return out.view(*lead, 4)

    def forward_raw(self, pts, viewdirs, feats, pe_layout=0):
        """Fused entry used by run_network_mvs: raw 3-vectors in, encodings computed inside the kernel.
        pts [N,S,3], viewdirs [N,3] or [N,S,3], feats [N,S,F] -> [N,S,4]."""
        N, S = pts.shape[0], pts.shape[1]
        flat, pw, ws = self._flat_and_stream(pe_layout)
        out = ops.mlp(flat, feats, pts, viewdirs, pw, S, ws)
        return out.view(N, S, 4)


class UCNeRF(nn.Module):
    """network/models.py:187-207."""

    def __init__(self, D=8, W=256, input_ch_pts=3, input_ch_views=3, input_ch_feat=8, skips=[4], net_type='v2',
                 fine=False, view_num=4):
        super().__init__()
        self.in_ch_pts, self.in_ch_views, self.in_ch_feat = input_ch_pts, input_ch_views, input_ch_feat
        self.nerf = BaseAdapt_Renderer(D=D, W=W, input_ch_feat=input_ch_feat, input_ch=input_ch_pts, output_ch=4,
                                       skips=skips, input_ch_views=input_ch_views, use_viewdirs=True, fine=fine,
                                       view_num=view_num)

    def forward_alpha(self, x):
        return self.nerf.forward_alpha(x)

    def forward_uncertainty(self, x):
        return 1 - x

    def forward(self, x):
        return self.nerf(x)

    def forward_raw(self, pts, viewdirs, feats, pe_layout=0):
        return self.nerf.forward_raw(pts, viewdirs, feats, pe_layout)


def create_ucnerf(args, pts_embedder=True, dir_embedder=True):
    """network/models.py:209-283: returns (render_kwargs_train, render_kwargs_test, start, grad_vars) with the same
    dict keys.  Differences forced by the environment and scope: device-agnostic (no .cuda() at import), and the
    consistency learner (CascadeMVSNet + its downloaded weights, out of scope) is taken from `args.network_mvs`
    if present; 'network_mvs' is None otherwise."""
    from .renderer import run_network_mvs
    if pts_embedder:
        embed_fn, input_ch = get_embedder(args.multires, args.i_embed)
    else:
        embed_fn, input_ch = None, args.pts_dim
    if dir_embedder:
        embeddirs_fn, input_ch_views = get_embedder(args.multires_views, args.i_embed)
    else:
        embeddirs_fn, input_ch_views = None, args.dir_dim
    dev = torch.device(getattr(args, "device", device))
    model = UCNeRF(D=args.netdepth, W=args.netwidth, input_ch_pts=input_ch, skips=[4], input_ch_views=input_ch_views,
                   input_ch_feat=args.feat_dim, net_type=args.net_type, view_num=args.view_num).to(dev)
    grad_vars = list(model.parameters())

    def network_query_fn(pts, viewdirs, rays_feats, network_fn):
        return run_network_mvs(pts, viewdirs, rays_feats, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                               netchunk=args.netchunk)

    encoding_net = getattr(args, "network_mvs", None)
    if encoding_net is not None and getattr(args, "finetune", None) is None:
        grad_vars += list(encoding_net.parameters())
    start = 0
    ckpt_path = getattr(args, "ckpt", None)
    if ckpt_path is not None and ckpt_path != 'None':
        print('Reloading from', ckpt_path)
        ckpt = torch.load(ckpt_path, map_location=dev)
        if encoding_net is not None:
            encoding_net.load_state_dict(ckpt['network_mvs_state_dict'])
        model.load_state_dict(ckpt['network_fn_state_dict'])
    render_kwargs_train = {
        'network_query_fn': network_query_fn, 'perturb': args.perturb, 'N_samples': args.N_samples, 'network_fn': model,
        'network_mvs': encoding_net, 'use_viewdirs': args.use_viewdirs, 'white_bkgd': args.white_bkgd,
        'raw_noise_std': args.raw_noise_std,
    }
    render_kwargs_test = dict(render_kwargs_train)
    render_kwargs_test['perturb'] = False
    return render_kwargs_train, render_kwargs_test, start, grad_vars
