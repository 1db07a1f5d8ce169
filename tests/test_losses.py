"""Row f3: the trainer's loss terms (train.py:164-188) against values and gradients captured from the reference's own
loss classes (tests/golden/make_golden.py g15).  CPU only."""
import torch

from conftest import load_golden
from uc_nerf_amd.utils import loss as L


def test_training_loss_terms_and_gradients_match_the_reference():
    g = load_golden("g15_losses")
    rgb = g["rgb"].clone().requires_grad_(True)
    depth = g["depth_pred"].clone().requires_grad_(True)
    outputs, gt, w = {}, {}, {}
    for k in ("stage1", "stage2", "stage3"):
        outputs[k] = {"depth": g[k + "_depth"].clone().requires_grad_(True)}
        gt[k], w[k] = g[k + "_gt"], g[k + "_w"]
    outputs["depth"] = outputs["stage3"]["depth"]
    loss, parts = L.training_loss(rgb, depth, g["target_s"], g["target_depths"], g["target_weights"], g["patch_dpt"], outputs, gt, w,
                                  n_rays=int(g["n_rays"]), patch_num=int(g["patch_num"]), patch_size=int(g["patch_size"]))
    close = lambda a, b: torch.testing.assert_close(a.detach(), torch.as_tensor(b, dtype=a.dtype).reshape(a.shape), atol=1e-7, rtol=1e-6)      # noqa: E731
    close(loss, g["loss"])
    for k in ("img_loss", "loss_mvs", "smooth_loss", "loss_nerf_depth", "loss_scaleinvariant"):
        close(parts[k], g[k])
    loss.backward()
    close(rgb.grad, g["g_rgb"]); close(depth.grad, g["g_depth"])
    for k in ("stage1", "stage2", "stage3"):
        close(outputs[k]["depth"].grad, g[k + "_g"])


def test_scale_and_shift_handles_a_degenerate_image():
    pred = torch.ones(2, 4, 4)
    pred[1] = torch.arange(16.0).view(4, 4)
    s, t = L.compute_scale_and_shift(pred, 2 * pred + 1, torch.ones_like(pred))
    assert s[0] == 0 and t[0] == 0                                       # singular normal equations -> zeros, as the reference
    torch.testing.assert_close(s[1], torch.tensor(2.0)); torch.testing.assert_close(t[1], torch.tensor(1.0), atol=1e-5, rtol=0)
