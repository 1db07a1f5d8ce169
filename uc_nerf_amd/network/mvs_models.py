"""Drop-in for the part of the reference's `network/mvs_models.py` that sits directly in front of the ray-marching
path: `DepthNet` (mvs_models.py:585-646).  Same call signature and result keys; the cost-volume assembly (`homo_warp`
+ mask count + variance) and the depth regression run as the HIP kernels `ucnerf_cost_volume` / `ucnerf_depth_regress`,
the 3D regularisation network stays the caller's `cost_regularization` module (MIOpen territory).

Both kernels have backward kernels behind `torch.autograd.Function`s, so the feature network trains through the
variance volume and the regularisation network through depth and photometric confidence, as in the reference.
`features` may be a list of [1,C,H,W] maps (as the reference passes) or a stacked tensor.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


def mvs_depth_regression(p, depth_values):                 # mvs_models.py:574-579
    if depth_values.dim() <= 2:
        depth_values = depth_values.view(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, 1)


class DepthNet(nn.Module):
    def forward(self, features, affine_mat_stage, affine_mat_inv_stage, depth_values, num_depth, cost_regularization, imgs,
                pad=0, prob_volume_init=None):
        features = torch.stack(list(features)) if not torch.is_tensor(features) else features      # [V,B,C,H,W]
        V, B, C, H, W = features.shape
        if B != 1:
            raise RuntimeError("uc_nerf_amd DepthNet: batch size 1 (as the reference's datasets provide)")
        if pad > 0:
            depth_values = F.pad(depth_values, (pad, pad, pad, pad), "replicate")
        # (src_proj @ ref_proj_inv)[:3] per source view (mvs_models.py:612); entry 0 of the stage matrices is the target view
        proj = (affine_mat_stage[1:V + 1] @ affine_mat_inv_stage[0:1])[:, :3].contiguous()
        variance = ops.cost_volume(features[:, 0], proj, depth_values[0], pad=pad)
        cost_feat_no_ref, prob = cost_regularization(variance.unsqueeze(0))
        prob_pre = prob.squeeze(1)
        prob_volume, depth, conf = ops.depth_regress(prob_pre[0], depth_values[0],
                                                     None if prob_volume_init is None else prob_volume_init[0], pad=pad)
        return {"depth": depth.unsqueeze(0), "photometric_confidence": conf.unsqueeze(0),
                "volume_feature_no_ref": cost_feat_no_ref, "depth_values": depth_values, "img_feats": features,
                "prob_volume": prob_volume.unsqueeze(0)}
