// a1 + a3 device arithmetic shared by rays.hip (the stand-alone ray / depth kernels) and mlp_bf16.hip (ABI v4: the gather-fused MLP launch of a
// coarse pass generates its own rays and stratified depths in the tile prologue).  One restatement, so both give the same bits:
// data/ray_utils.py:27,32-53 (pinhole directions, rays_d = d_cam @ R^T), network/renderer.py:232-238 (view-direction feature),
// data/ray_utils.py:176-194 (linspace depths, stratified jitter).  Compile with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace ucnerf {

// intrinsic-matrix (+z) convention: d_cam = ((x - K02) / K00, (y - K12) / K11, 1), rays_d = d_cam @ R^T (R = c2w[:, :3], row-major with 4 columns)
__device__ __forceinline__ void pinhole_ray(float x, float y, float k00, float k02, float k11, float k12, const float* R, float* wx, float* wy, float* wz) {
    const float dx = (x - k02) / k00, dy = (y - k12) / k11, dz = 1.0f;
    *wx = dx * R[0] + dy * R[1] + dz * R[2];
    *wy = dx * R[4] + dy * R[5] + dz * R[6];
    *wz = dx * R[8] + dy * R[9] + dz * R[10];
}

// angle = (d / |d|) @ Q^T (Q row-major with 4 columns): the arithmetic of dir_feature_kernel
__device__ __forceinline__ void view_dir_feature(float wx, float wy, float wz, const float* Q, float* ax, float* ay, float* az) {
    const float c = sqrtf(wx * wx + wy * wy + wz * wz);
    const float ux = wx / c, uy = wy / c, uz = wz / c;
    *ax = ux * Q[0] + uy * Q[1] + uz * Q[2];
    *ay = ux * Q[4] + uy * Q[5] + uz * Q[6];
    *az = ux * Q[8] + uy * Q[9] + uz * Q[10];
}

// torch.linspace(0, 1, S)[i]: ATen computes start + step*i below the midpoint and end - step*(S-1-i) above.
__device__ __forceinline__ float linspace01(int i, int S) {
    if (S == 1) return 0.f;
    float step = 1.0f / (float)(S - 1);
    return i < S / 2 ? step * (float)i : 1.0f - step * (float)(S - 1 - i);
}

__device__ __forceinline__ float z_at(float near, float far, int i, int S, int lindisp) {
    float t = linspace01(i, S);
    return lindisp ? 1.f / (1.f / near * (1.f - t) + 1.f / far * t) : near * (1.f - t) + far * t;
}

// depth s of S on [near, far]; perturb > 0: z = lower + (upper - lower) * (perturb * noise) between the mid-points to the neighbours
__device__ __forceinline__ float stratified_depth(float near, float far, int s, int S, int lindisp, float perturb, float noise) {
    float z = z_at(near, far, s, S, lindisp);
    if (perturb > 0.f) {
        float zl = s > 0 ? z_at(near, far, s - 1, S, lindisp) : z;
        float zu = s + 1 < S ? z_at(near, far, s + 1, S, lindisp) : z;
        float lower = s > 0 ? .5f * (zl + z) : z;         // mids = .5*(z[:-1] + z[1:])
        float upper = s + 1 < S ? .5f * (z + zu) : z;
        z = lower + (upper - lower) * (perturb * noise);
    }
    return z;
}

}  // namespace ucnerf
