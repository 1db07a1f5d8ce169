// Packed weight-stream layout of the UC-NeRF MLP for the gfx950 MFMA kernels (host + device shared).
//
// The MLP (network/models.py:138-184 of the reference) is evaluated TRANSPOSED: Y^T[feature][sample] =
// W[feature][k] * X^T[k][sample] with v_mfma_f32_32x32x2_f32, so that
//   - the weights are the A operand: lane l supplies W[32*nt + (l&31)][k(l>>5)] -- one float per MFMA;
//   - a 32-sample tile of activations is the B operand AND the accumulator: lane l holds sample (l&31),
//     and its 16 accumulator registers of row-tile nt hold features 32*nt + (r&3) + 8*(r>>2) + 4*(l>>5).
// An accumulator register can therefore be fed back as the B operand of the next layer's k-step without
// any data movement: register r of tile kt pairs input features f(kt,r,0) [lanes 0-31] and f(kt,r,1)
// [lanes 32-63].  The weight stream is packed in exactly the order the kernel walks it: per k-step one
// float4 per lane = the A operands of the four row-tiles nt = 0..3.
#pragma once
#include <cstdint>

namespace ucnerf {

constexpr int MLP_W = 128;        // hidden width
constexpr int MLP_PE_PTS = 63;    // 3 + 6*10
constexpr int MLP_PE_DIR = 27;    // 3 + 6*4
constexpr int KS_PE_PTS = 32;     // k-steps of the padded 64-wide point encoding
constexpr int KS_PE_DIR = 16;     // k-steps of the direction encoding (28 wide, padded to 32: sections are multiples of 4 k-steps)
constexpr int KS_HID = 64;        // k-steps of a 128-wide hidden input
constexpr int KSTEP_FLOATS = 256; // 64 lanes x 4 row-tiles
#ifndef UCNERF_MLP_RING
#define UCNERF_MLP_RING 4
#endif
constexpr int RING = UCNERF_MLP_RING;   // k-steps of weight prefetch kept in flight per wave (sections are multiples of it)
constexpr int N_SEC = 10;         // GEMM sections: bd, L0..L5, bc, ft, vc
constexpr int CONST_FLOATS = N_SEC * 128 + 2 * 516;   // bias blocks + two head blocks (per-block LDS copy)

enum { SEC_BD = 0, SEC_L0 = 1, SEC_BC = 7, SEC_FT = 8, SEC_VC = 9 };

// feature index held by accumulator register r (0..15) of row-tile kt on lane-half h
__host__ __device__ inline int acc_feature(int kt, int r, int h) { return 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h; }

// Stream = [k-steps of all sections in execution order][wrap: copy of the first RING k-steps][constants].
//   execution order: bd | L0 | L1..L4 | L5 (pe part, hidden part) | bc | ft | vc (hidden part, dir part)
//   constants: N_SEC bias blocks [h][nt][r] (128 floats each), then head blocks hb, ha ([h][nt][r][4] + 4 biases).
struct MlpLayout {
    int v;            // source views
    int F;            // feature row length 24 + 12 v + 1
    int kd, kc;       // k-steps of the two bias nets, rounded up to a multiple of RING
    int64_t ks_total; // k-steps per tile
    int64_t off_sec[N_SEC];   // float offset of each section's first k-step
    int64_t off_wrap, off_const, total;
    // flat-parameter offsets (floats), state_dict order
    int64_t p_lw[6], p_lb[6], p_bdw, p_bdb, p_bcw, p_bcb, p_c1w, p_c1b, p_vw, p_vb, p_vcw, p_vcb, p_fw, p_fb,
        p_f1w, p_f1b, p_clw, p_clb, p_aw, p_ab, p_a1w, p_a1b, p_rw, p_rb, p_crw, p_crb, n_params;
};

inline bool mlp_layout(int v, MlpLayout* L) {
    if (v < 1 || v > 8) return false;
    L->v = v;
    L->F = 24 + 12 * v + 1;
    L->kd = ((24 + 4 * v) / 2 + RING - 1) / RING * RING;
    L->kc = (4 * v + RING - 1) / RING * RING;
    const int ks[N_SEC] = {L->kd, KS_PE_PTS, KS_HID, KS_HID, KS_HID, KS_HID, KS_PE_PTS + KS_HID, L->kc, KS_HID,
                           KS_HID + KS_PE_DIR};
    int64_t o = 0;
    for (int i = 0; i < N_SEC; ++i) { L->off_sec[i] = o; o += (int64_t)ks[i] * KSTEP_FLOATS; }
    L->ks_total = o / KSTEP_FLOATS;
    L->off_wrap = o;  o += (int64_t)RING * KSTEP_FLOATS;
    L->off_const = o; o += CONST_FLOATS;
    L->total = o;
    // flat parameters
    int64_t q = 0;
    auto take = [&](int64_t n) { int64_t r = q; q += n; return r; };
    const int W = MLP_W;
    for (int i = 0; i < 6; ++i) {
        int k = i == 0 ? MLP_PE_PTS : (i == 5 ? W + MLP_PE_PTS : W);
        L->p_lw[i] = take((int64_t)W * k);
        L->p_lb[i] = take(W);
    }
    L->p_bdw = take((int64_t)W * (24 + 4 * v)); L->p_bdb = take(W);
    L->p_bcw = take((int64_t)W * 8 * v);        L->p_bcb = take(W);
    L->p_c1w = take(1);                         L->p_c1b = take(1);
    L->p_vw = take((int64_t)(W / 2) * (MLP_PE_DIR + W));  L->p_vb = take(W / 2);
    L->p_vcw = take((int64_t)(W / 2) * (MLP_PE_DIR + W)); L->p_vcb = take(W / 2);
    L->p_fw = take((int64_t)W * W);  L->p_fb = take(W);
    L->p_f1w = take((int64_t)W * W); L->p_f1b = take(W);
    L->p_clw = take((int64_t)W * W); L->p_clb = take(W);
    L->p_aw = take(W / 2);  L->p_ab = take(1);
    L->p_a1w = take(W);     L->p_a1b = take(1);
    L->p_rw = take(3 * (W / 2)); L->p_rb = take(3);
    L->p_crw = take(3 * W);      L->p_crb = take(3);
    L->n_params = q;
    return true;
}

// Activations the training forward keeps for the backward: [m,128] each, row-major -- float32 (p24 = 0: what the layer-by-layer backward,
// bwd_mode 1, reads) or the 24-bit format of p24.h (p24 = 1: the gradient chain and the weight-gradient launch, bwd_mode 0).
struct MlpSaved {
    int p24;
    float* bd;      // depth-bias net output b_d
    float* h[6];    // trunk activations h_l = relu((W_l x + b_l) * b_d)
    float* bc;      // confidence-bias net output b_c
    float* ft;      // feature_linear output f
    float* vc;      // relu([views_linears | view_confi_linears]([f | dir encoding]))
};

// Encoding argument handled by k-step t of lane-half h: a in [0, 3*n_freqs) = 3*freq + coord; returns
// kind 0 = sin(arg a), 1 = cos(arg a), 2 = raw coordinate `a`, 3 = zero pad.
__host__ __device__ inline void pe_slot(int t, int h, int n_freqs, int* kind, int* a) {
    int half = 3 * n_freqs / 2;
    if (t < half) { *kind = 0; *a = 2 * t + h; }
    else if (t < 2 * half) { *kind = 1; *a = 2 * (t - half) + h; }
    else if (t == 2 * half) { *kind = 2; *a = h ? 2 : 0; }
    else if (t == 2 * half + 1) { *kind = h ? 3 : 2; *a = 1; }
    else { *kind = 3; *a = 0; }
}

// column of the reference's encoded vector for a slot (layout 0 = network/models.py, 1 = run_nerf_helpers.py)
__host__ __device__ inline int pe_column(int kind, int a, int n_freqs, int layout) {
    if (kind == 2) return a;
    if (kind == 3) return -1;
    int fr = a / 3, c = a % 3;
    if (layout == 0) return 3 + (kind == 0 ? 0 : 3 * n_freqs) + a;
    return 3 + 6 * fr + (kind == 0 ? 0 : 3) + c;
}

}  // namespace ucnerf
