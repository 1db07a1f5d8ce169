"""Import harness for the UC-NeRF reference (only usable where /root/reference exists).

TEST INFRASTRUCTURE ONLY.  Used by make_golden.py (fixture generation) and by the optional
`tests/test_oracle_vs_reference.py` cross-check.  Never imported by the product package, never
available on the GPU box (the reference tree does not travel).

The reference imports a few third-party modules at module scope that are absent from this
image (cv2, torchvision.transforms, kornia, warmup_scheduler, tkinter, inplace_abn).  None of
them is executed on the ray-marching hot path, so inert stand-in modules are registered in
``sys.modules`` before import; `Tensor.cuda` is made a no-op because
network/models.py:40 calls `.cuda()` unconditionally and there is no GPU here.

One exception, for the cost-volume fixtures (G12): ``homo_warp`` (utils/utils.py:1105) calls
``kornia.utils.create_meshgrid`` (kornia >= 0.6.12, requirements.txt:12).  That one function is supplied here,
restated from its published behaviour (pixel-coordinate grid [1,H,W,2], last dim (x, y)); everything else of kornia
stays inert.
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get("UCNERF_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "network"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load():
    """Returns a namespace with the reference hot-path modules."""
    import torch

    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)

    class _Inert:  # placeholder class for names that are only referenced, never called
        def __init__(self, *a, **k):
            raise RuntimeError("inert stand-in called")

    for name in ("cv2", "warmup_scheduler", "tkinter", "inplace_abn", "kornia", "kornia.utils",
                 "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            _stub(name)
    sys.modules["cv2"].COLORMAP_MAGMA = 0
    sys.modules["cv2"].COLORMAP_JET = 0
    sys.modules["tkinter"].X = None
    sys.modules["inplace_abn"].InPlaceABN = _Inert
    sys.modules["warmup_scheduler"].GradualWarmupScheduler = _Inert
    def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
        xs = torch.linspace(0, width - 1, width, device=device, dtype=dtype)
        ys = torch.linspace(0, height - 1, height, device=device, dtype=dtype)
        if normalized_coordinates:
            xs = (xs / (width - 1) - 0.5) * 2
            ys = (ys / (height - 1) - 0.5) * 2
        base = torch.stack(torch.meshgrid([xs, ys], indexing="ij"), dim=-1)      # W x H x 2
        return base.permute(1, 0, 2).unsqueeze(0)                                  # 1 x H x W x 2

    sys.modules["kornia"].create_meshgrid = create_meshgrid
    sys.modules["kornia.utils"].create_meshgrid = create_meshgrid
    sys.modules["kornia"].utils = sys.modules["kornia.utils"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self

    if REF not in sys.path:
        sys.path.insert(0, REF)
    # `data/__init__.py` pulls in the datasets (imageio/cv2 IO) -> load ray_utils by path instead.
    import utils.utils as ref_utils            # noqa: E402
    import utils.run_nerf_helpers as ref_helpers  # noqa: E402
    import network.renderer as ref_renderer    # noqa: E402
    import network.models as ref_models        # noqa: E402
    spec = importlib.util.spec_from_file_location("ref_ray_utils", os.path.join(REF, "data", "ray_utils.py"))
    ref_ray_utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_ray_utils)
    import network.mvs_models as ref_mvs       # noqa: E402  (DepthNet / homo_warp call site, for the cost-volume fixtures)
    ref_utils.create_meshgrid = create_meshgrid   # utils/utils.py:1102 binds the name at import time
    torch.autograd.set_detect_anomaly(False)   # the reference switches it on at import
    return types.SimpleNamespace(utils=ref_utils, helpers=ref_helpers, renderer=ref_renderer,
                                 models=ref_models, ray_utils=ref_ray_utils, mvs=ref_mvs)
