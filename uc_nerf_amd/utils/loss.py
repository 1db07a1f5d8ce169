"""Drop-in for the losses the reference's training step applies to the rendered batch:
`utils/loss.py` (`img2mse`, `GradientLoss` with its scale-and-shift alignment), `EdgePreservingSmoothnessLoss` and
`cas_mvsnet_loss` (`network/mvs_models.py:491-529`), plus the weighting of `train.py:164-188` as one function.

These are reductions over at most a few thousand values per step; they are plain torch expressions on the tensors' own
device (autograd included), not kernels.  Same names, argument orders and results as the reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

img2mse = lambda x, y: torch.mean((x - y) ** 2)                      # noqa: E731  (utils/loss.py:3)
img2mse_weight = lambda x, y, w: torch.mean((x - y) ** 2 * w)        # noqa: E731  (utils/loss.py:4)


def compute_scale_and_shift(prediction, target, mask):
    """Least-squares scale and shift per image, closed form of the 2x2 normal equations (utils/loss.py:53-77)."""
    a00 = torch.sum(mask * prediction * prediction, (1, 2))
    a01 = torch.sum(mask * prediction, (1, 2))
    a11 = torch.sum(mask, (1, 2))
    b0 = torch.sum(mask * prediction * target, (1, 2))
    b1 = torch.sum(mask * target, (1, 2))
    det = a00 * a11 - a01 * a01
    ok = det != 0
    safe = torch.where(ok, det, torch.ones_like(det))
    x0 = torch.where(ok, (a11 * b0 - a01 * b1) / safe, torch.zeros_like(b0))
    x1 = torch.where(ok, (-a01 * b0 + a00 * b1) / safe, torch.zeros_like(b1))
    return x0, x1


def gradient_loss(prediction, target, mask=None, reduction=None):
    """utils/loss.py:79-95: summed absolute first differences of the residual, averaged over the images."""
    diff = prediction - target
    gx = torch.abs(diff[:, :, 1:] - diff[:, :, :-1])
    gy = torch.abs(diff[:, 1:, :] - diff[:, :-1, :])
    return torch.sum(torch.sum(gx, (1, 2)) + torch.sum(gy, (1, 2))) / prediction.shape[0]


class GradientLoss(nn.Module):
    """utils/loss.py:97-113 (the trainer's `edge_loss`)."""

    def __init__(self, scales=4, reduction='batch-based'):
        super().__init__()

    def forward(self, prediction, target, mask):
        scale, shift = compute_scale_and_shift(prediction, target, mask)
        return gradient_loss(scale.view(-1, 1, 1) * prediction + shift.view(-1, 1, 1), target, mask)


class EdgePreservingSmoothnessLoss(nn.Module):
    """network/mvs_models.py:491-508 (the trainer's `smooth_loss`): bilateral weights from the prior, four neighbour directions."""

    def __init__(self):
        super().__init__()
        self.patch_size, self.gamma = 4, 0.1

    # the four neighbour directions as (row shift, column shift) pairs: right, down, down-right, up-right
    _NEIGHBOURS = ((0, 1), (1, 0), (1, 1), (-1, 1))

    @staticmethod
    def _pair(x, dr, dc):
        """x and its (dr, dc) neighbour, cropped to the overlap, for [n, rows, cols, ...] tensors."""
        rows, cols = x.shape[1], x.shape[2]
        r0, r1 = (0, rows - dr) if dr >= 0 else (-dr, rows)
        return x[:, r0:r1, 0:cols - dc], x[:, r0 + dr:r1 + dr, dc:cols]

    def forward(self, inputs, weights):
        total = 0
        for dr, dc in self._NEIGHBOURS:
            wa, wb = self._pair(weights, dr, dc)
            xa, xb = self._pair(inputs, dr, dc)
            bilateral = torch.exp(-torch.abs(wa - wb).sum(-1) / self.gamma)
            total = total + torch.mean(torch.abs(bilateral * (xa - xb)))
        return total / 4


def cas_mvsnet_loss(inputs, depth_gt_ms, weight_ms, with_weight=True, mvs_type=0, **kwargs):
    """network/mvs_models.py:507-529: smooth-L1 on the valid sparse depths of every cascade stage, weights 0.5 / 1 / 2.
    (The ground truth is moved to the estimate's device instead of `.cuda()`.)"""
    stage_w = [0.5, 1.0, 2.0]
    total, depth_loss = 0, None
    for key in [k for k in inputs.keys() if "stage" in k]:
        est = inputs[key]["depth"]
        gt = depth_gt_ms[key].to(est.device)
        w = weight_ms[key].to(est.device)
        w = w[w > 0]
        mask = gt > 0
        if with_weight:
            depth_loss = (F.smooth_l1_loss(est[mask], gt[mask], reduction='none') * w).mean()
        else:
            depth_loss = F.smooth_l1_loss(est[mask], gt[mask], reduction='mean')
        total = total + stage_w[int(key.replace("stage", "")) - 1] * depth_loss
    return total, depth_loss


def training_loss(rgb, depth_pred, target_s, target_depths, target_weights, patch_dpt, mvs_outputs, depth_sparse_ms, weight_ms,
                  n_rays, patch_num, patch_size, smooth_loss=None, edge_loss=None):
    """The loss of `train.py:164-188`.  The batch is [patch rays (patch_num * patch_size^2) | random rays | rays_depth rays
    from `n_rays` on]; `patch_dpt` is the DPT prior at the patch pixels [patch_num, patch_size, patch_size, 1].
    Returns (loss, parts)."""
    smooth_loss = smooth_loss or EdgePreservingSmoothnessLoss()
    edge_loss = edge_loss or GradientLoss()
    patch_pts = patch_num * patch_size * patch_size
    loss_mvs, _ = cas_mvsnet_loss(mvs_outputs, depth_sparse_ms, weight_ms)
    patch_depth = depth_pred[:patch_pts].reshape(-1, patch_size, patch_size)
    smooth = smooth_loss(patch_depth[:patch_num // 2, ...], patch_dpt[:patch_num // 2, ...])
    loss_nerf_depth = torch.mean(((depth_pred[n_rays:] - target_depths) ** 2) * target_weights)
    scale_inv = edge_loss(patch_depth[patch_num // 2:, ...], patch_dpt[patch_num // 2:, ...].squeeze(-1),
                          torch.ones_like(patch_depth[patch_num // 2:, ...]))
    depth_loss = loss_nerf_depth * 0.05 + loss_mvs * 0.05 + smooth * 0.05 + scale_inv * 0.008
    img_loss = img2mse(rgb, target_s)
    loss = depth_loss + img_loss * 5.0
    return loss, {"img_loss": img_loss, "loss_nerf_depth": loss_nerf_depth, "loss_mvs": loss_mvs, "smooth_loss": smooth,
                  "loss_scaleinvariant": scale_inv}
