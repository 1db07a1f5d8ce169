// Interfaces between the translation units of the MLP backward (bwd_mode 0, include/ucnerf_hip.h):
//   mlp_bwd.hip        host orchestration (ucnerf_mlp_bwd) + the layer-by-layer kernels of bwd_mode 1
//   mlp_bwd_chain.hip  every data gradient of a 32-sample tile in one register-resident kernel
//   mlp_wgrad.hip      every weight-gradient product in one persistent launch
#pragma once
#include "common.h"
#include "mlp_layout.h"
#include "p24.h"

namespace ucnerf {

// ---- mlp_bwd_chain.hip
size_t bwd_chain_stream_floats();
int launch_pack_bwd(int n_src, const float* flat, float* stream_out, hipStream_t st);
int launch_mlp_bwd_chain(int n_src, int m, const float* raw, const float* g_raw, const float* feats, int ldf, int feats_tiled, const MlpSaved* sv, const float* stream,
                         float* G_vc, float* G_f, float* G_bc, float* gx, float* G_bd, float* const* G_y, float* g_feats, int ldgf, float* g_base,
                         float* g_adapt, float* g_sigma, hipStream_t st);

// ---- mlp_wgrad.hip: gW[n][k] += sum_s G[s][n] X[s / xdiv][k], gb[n] += sum_s G[s][n] for a list of (G, X) pairs
constexpr int WG_MAX_PAIRS = 16;
struct WgPair {
    const void* G;                        // [m, nout] in the 24-bit format (p24.h), rows ldg BYTES apart (the gradient chain writes every G)
    const void* X;                        // xmode 0: [m, w] in the 24-bit format;  1: fp32 [m / xdiv, w];  2: fp32 in the MLP tile layout (xtile_f); rows ldx BYTES apart
    float* gW; float* gb;                 // gW[n][k] at gW + n * ldw + k; gb may be NULL
    float* gW_hi; float* gb_hi;           // optional: rows n >= split belong to a second layer (row n - split of these)
    int ldg, nout, ldx, w, xdiv, ldw, split, cost;
    int xmode;
    int xtile_f;                          // xmode 2: X lives in the MLP tile layout [m / 32][xtile_f][32] (X points at its first column's row): ldx unused
    unsigned div_m, div_sh;               // s / xdiv = __umulhi(s, div_m) >> div_sh
};
struct WgArgs {
    WgPair p[WG_MAX_PAIRS];
    long long prefix[WG_MAX_PAIRS + 1];   // cumulative cost: pair i covers [prefix[i], prefix[i + 1]) = stages * cost_i
    int n_pairs, m, stages;
    int chunk;                            // stages per chunk of the hand-out (set by wgrad_launch)
    unsigned* counters;                   // WG_MAX_PAIRS counters in global memory, zeroed by wgrad_launch: the next chunk of every pair (mlp_wgrad.hip)
};
void wgrad_begin(WgArgs* a, int m, unsigned* counters);
int wgrad_add(WgArgs* a, const void* G, int ldg_bytes, int nout, const void* X, int x24, int ldx, int xdiv, int w, float* gW, int ldw, float* gb, float* gW_hi,
              float* gb_hi, int split, int xtile_f = 0);
int wgrad_launch(const WgArgs* a, hipStream_t st);

}  // namespace ucnerf
