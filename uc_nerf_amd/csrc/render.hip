// a10: one render pass = projection -> feature gather -> PE + MLP -> composite, chained on one stream
// (network/renderer.py:215-255 with utils/utils.py:716-724 in front).  Host-side orchestration only:
// every stage is one of the library's own kernels; intermediates live in the caller's workspace.
#include "common.h"
#include <atomic>
#include <cstdlib>
#include <mutex>
#include "mlp_layout.h"

namespace ucnerf {

struct PointsArgs {
    int n, S;
    const float* rays_o;
    const float* rays_d;
    const float* z;
    float w2c[12];
    float K[9];
    float inv_w, inv_h;        // W-1, H-1
    float near, far;
    const float* near_far;     // [n,6] or NULL
    float* pts;
    float* ndc1;
    float* ndc2;
    float* ndc3;
    float* ndc;
};

// world point of every sample + its four normalised copies (utils/utils.py:716, :333-367)
__global__ void __launch_bounds__(256) render_points_kernel(PointsArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.n * a.S) return;
    const int r = (int)(idx / a.S);
    const float z = a.z[idx];
    const float x = a.rays_o[0] + z * a.rays_d[3 * r], y = a.rays_o[1] + z * a.rays_d[3 * r + 1],
                w = a.rays_o[2] + z * a.rays_d[3 * r + 2];
    const size_t o = 3 * (size_t)idx;
    a.pts[o] = x; a.pts[o + 1] = y; a.pts[o + 2] = w;
    const float* M = a.w2c;
    const float cx = x * M[0] + y * M[1] + w * M[2] + M[3];
    const float cy = x * M[4] + y * M[5] + w * M[6] + M[7];
    float cz = x * M[8] + y * M[9] + w * M[10] + M[11];
    if (fabsf(cz) < 1e-4f) cz = 1e-4f;
    const float* K = a.K;
    const float qx = cx * K[0] + cy * K[1] + cz * K[2];
    const float qy = cx * K[3] + cy * K[4] + cz * K[5];
    const float qz = cx * K[6] + cy * K[7] + cz * K[8];
    const float u = (qx / qz + 0.0f) / a.inv_w, v = (qy / qz + 0.0f) / a.inv_h;
    float n1 = a.near, f1 = a.far, n2 = a.near, f2 = a.far, n3 = a.near, f3 = a.far;
    if (a.near_far) {
        const float* nf = a.near_far + 6 * (size_t)r;
        n1 = nf[0]; f1 = nf[1]; n2 = nf[2]; f2 = nf[3]; n3 = nf[4]; f3 = nf[5];
    }
    a.ndc1[o] = u; a.ndc1[o + 1] = v; a.ndc1[o + 2] = (qz - n1) / (f1 - n1);
    a.ndc2[o] = u; a.ndc2[o + 1] = v; a.ndc2[o + 2] = (qz - n2) / (f2 - n2);
    a.ndc3[o] = u; a.ndc3[o + 1] = v; a.ndc3[o + 2] = (qz - n3) / (f3 - n3);
    a.ndc[o] = u; a.ndc[o + 1] = v; a.ndc[o + 2] = (qz - a.near) / (a.far - a.near);
}

int launch_gather_cl(const ucnerf_render_params* p, float* feats, int tiled, float* ndc, hipStream_t st);
int check_cl_sources(const ucnerf_render_params* p, const char* who);                                              // gather_cl.hip
int launch_mlp_fwd_bf16x3_gather(const ucnerf_render_params* rp, const float* dirs, float* raw, hipStream_t st,
                                 const ucnerf_composite_params* tail_c, const ucnerf_sample_pdf_params* tail_s, float* tail_dir_out);   // mlp_bf16.hip

struct Workspace {
    float *pts, *ndc1, *ndc2, *ndc3, *ndc, *angle, *feats, *raw;
};

static inline size_t align4(size_t x) { return (x + 3) & ~(size_t)3; }

static size_t carve(float* base, int n, int S, int V, Workspace* w) {
    const size_t M = (size_t)n * S, F = 24 + 12 * V + 1;
    const size_t tiles = (M + 31) / 32;
    size_t o = 0;
    auto take = [&](size_t k) { float* r = base ? base + o : nullptr; o += align4(k); return r; };
    w->pts = take(3 * M); w->ndc1 = take(3 * M); w->ndc2 = take(3 * M); w->ndc3 = take(3 * M); w->ndc = take(3 * M);
    w->angle = take(3 * M);          // view directions in the reference camera frame, one per RAY (the MLP kernels index them by sample / S)
    w->raw = take(4 * M);
    w->feats = take(tiles * 32 * F);
    return o;
}

static int launch_points(const ucnerf_render_params* p, hipStream_t st, Workspace* w) {
    const long long M = (long long)p->n * p->S;
    PointsArgs a;
    a.n = p->n; a.S = p->S; a.rays_o = p->rays_o; a.rays_d = p->rays_d; a.z = p->z;
    memcpy(a.w2c, p->w2c_ref, sizeof(a.w2c));
    memcpy(a.K, p->K_ref, sizeof(a.K));
    a.inv_w = (float)(p->W - 1); a.inv_h = (float)(p->H - 1);
    a.near = p->near; a.far = p->far; a.near_far = p->near_far;
    a.pts = w->pts; a.ndc1 = w->ndc1; a.ndc2 = w->ndc2; a.ndc3 = w->ndc3; a.ndc = w->ndc;
    hipLaunchKernelGGL(render_points_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, a);
    return check_launch("render_points");
}

// coordinates handed over by the caller (rendering() of the reference receives rays_pts / rays_ndc) instead of derived here
static inline bool coords_given(const ucnerf_render_params* p) { return p->pts_in != nullptr; }
// channel-last sources handed over / built (ABI v5: all five arrays or none)
static inline bool cl_given(const ucnerf_render_params* p) { return p->cl.vol[0] || p->cl.vol[1] || p->cl.vol[2] || p->cl.img_feat || p->cl.imgs; }

static void gather_geometry(const ucnerf_render_params* p, const Workspace* w, ucnerf_feat_gather_params* g) {
    memset(g, 0, sizeof(*g));
    g->m = p->n * p->S; g->V = p->cfg.n_src; g->H = p->H; g->W = p->W;
    for (int k = 0; k < 3; ++k) { g->vol_d[k] = p->vol_d[k]; g->vol_h[k] = p->vol_h[k]; g->vol_w[k] = p->vol_w[k]; g->vol[k] = p->vol[k]; }
    if (coords_given(p)) { g->pts = p->pts_in; g->ndc1 = p->ndc1_in; g->ndc2 = p->ndc2_in; g->ndc3 = p->ndc3_in; }
    else { g->pts = w->pts; g->ndc1 = w->ndc1; g->ndc2 = w->ndc2; g->ndc3 = w->ndc3; }
    g->conf = p->conf; g->imgs = p->imgs; g->img_feat = p->img_feat; g->w2cs = p->w2cs; g->intrinsics = p->intrinsics;
}

static void mlp_args(const ucnerf_render_params* p, const Workspace* w, const float* feats, int tiled, float* raw, ucnerf_mlp_params* m) {
    memset(m, 0, sizeof(*m));
    m->cfg = p->cfg; m->m = p->n * p->S; m->S = p->S; m->dirs_per_sample = 0; m->feats_tiled = tiled; m->max_blocks = p->max_blocks;
    m->pts = coords_given(p) ? p->ndc_in : w->ndc; m->dirs = w->angle; m->feats = feats; m->wstream = p->wstream; m->raw = raw;
}

static int launch_dirs(const ucnerf_render_params* p, hipStream_t st, Workspace* w) {
    ucnerf_dir_feature_params d;
    memset(&d, 0, sizeof(d));
    d.n = p->n; d.has_ref = 1; d.repeat = 1; memcpy(d.w2c_ref, p->w2c_dir, sizeof(d.w2c_ref));
    d.rays_d = p->rays_d; d.angle = w->angle; d.cos_angle = nullptr;
    d.w2c_ref_dev = p->w2c_dir_dev;                     // (the rotation on the device, when the caller holds it there: rendering()'s pose_ref['w2cs'][0])
    return ucnerf_dir_feature(&d, st);
}

static size_t carve_bwd_render(float* base, int n, int S, int V, Workspace* w, float** g_raw, float** g_feats, float** mlp_ws) {
    size_t o = carve(base, n, S, V, w);
    const size_t M = (size_t)n * S, F = 24 + 12 * V + 1;
    ucnerf_mlp_config cfg{V, 0, 0};
    const size_t mlp = (size_t)ucnerf_mlp_bwd_workspace_floats(&cfg, (int)M);
    auto take = [&](size_t k) { float* r = base ? base + o : nullptr; o += align4(k); return r; };
    *g_raw = take(4 * M); *g_feats = take(M * F); *mlp_ws = take(mlp);
    return o;
}

static std::atomic<long long> g_tail_launches{0};

// compositing (K7) parameters of the pass, `raw` = where the MLP leaves its outputs
static void composite_args(const ucnerf_render_params* p, const float* raw, ucnerf_composite_params* c) {
    memset(c, 0, sizeof(*c));
    c->n = p->n; c->S = p->S; c->variant = 0; c->white_bkgd = p->white_bkgd;
    c->raw = raw; c->z = p->z;
    c->rgb_map = p->rgb_map; c->depth_map = p->depth_map; c->acc_map = p->acc_map; c->weights = p->weights; c->var = p->var;
    c->u = p->u_sampled; c->wu = p->wu_map;
}

// May the gather-fused launch composite (and re-sample) its rays in its own tail?  Passes of at most three rounds of tiles (the shards of a strongly
// scaled batch: there the two or three latency-bound launches behind the MLP are a tenth of the step), whole 32-sample tiles per ray, the
// stand-alone kernels' small LDS shapes.  UCNERF_FUSED_TAIL=0 in the environment switches it off (A/B, tests: the outputs are bit-identical).
// The route's switch and size limit: atomics, initialised ONCE from the environment under std::call_once (several host threads may drive the
// library -- round 4's advisor finding); ucnerf_set_fused_tail() flips the switch at run time (A/B scripts, the bit-identity tests).
static std::atomic<int> g_tail_on{1}, g_tail_half_rounds{6};
static void tail_knobs_init() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = getenv("UCNERF_FUSED_TAIL");
        if (e && e[0] == '0') g_tail_on.store(0);
        const char* hr = getenv("UCNERF_FUSED_TAIL_HALF_ROUNDS");
        if (hr && atoi(hr) > 0) g_tail_half_rounds.store(atoi(hr));
    });
}
static bool tail_size_fits(int n, int S) {
    tail_knobs_init();
    if (!g_tail_on.load(std::memory_order_relaxed)) return false;
    const int cus = device_cus();
    if (cus <= 0 || n < 1 || S < 1 || S > 256) return false;      // (round 5: any S -- a block's tiles start at its own first sample)
    // the largest pass, in half rounds of tiles (a round = CUs x 8 tiles): three rounds.  Measured with the rays generated in the prologue (same box, 64 + 128):
    // 512 rays -4.5 %, 1024 rays -1.1 % (fine pass = 3 rounds), 2048 rays -0.2 .. -0.6 % (up to 6 rounds), 4096 rays +0.3 % (4 + 12 rounds): the
    // whole-rays-per-block dealing costs large passes what the folded launches save them.  UCNERF_FUSED_TAIL_HALF_ROUNDS overrides (tuning).
    // whole rays per block: rays / ceil(rays / CUs) blocks -- all but a tenth of the CUs must get one (37 rays: + 13 % on 37 blocks)
    const int rpb = cdiv(n, cus), blocks = cdiv(n, rpb);
    if ((long long)cdiv(rpb * S, 32) * 2 > (long long)g_tail_half_rounds.load(std::memory_order_relaxed) * 8) return false;      // tiles per block, in half rounds of eight
    return blocks * 10 >= cus * 9;
}
// the re-sampling a tail can do itself (the stand-alone kernels' small LDS shapes)
static bool tail_resample_fits(int S, int n_samples, int u_stride) {
    return S >= 3 && S - 1 <= 128 && S + n_samples <= 512 && n_samples >= 1 && (u_stride == 0 || u_stride == n_samples);
}
static bool tail_fits(const ucnerf_render_params* p, const ucnerf_sample_pdf_params* s) {
    if (!tail_size_fits(p->n, p->S) || p->max_blocks > 0 || p->cl.bf16) return false;      // (round 5: given coordinates take the route too)
    if (s && !(s->from_coarse && s->n == p->n && s->n_merge == p->S && s->n_bins == p->S - 1 && tail_resample_fits(p->S, s->n_samples, s->u_stride) &&
               (s->samples || s->inds || s->cdf || s->z_sorted) && (!s->merge_rank || s->z_sorted) && s->u)) return false;
    return true;
}

static int run_forward(const ucnerf_render_params* p, hipStream_t st, Workspace* w) {
    const int V = p->cfg.n_src;
    carve(p->workspace, p->n, p->S, V, w);
    int rc;
    const bool keep_feats = p->feats != nullptr;        // caller wants row-major features (for the backward)
    ucnerf_feat_gather_params g;
    float* raw_fused = nullptr;
    ucnerf_composite_params c;
    ucnerf_sample_pdf_params s_res;
    if (p->resample) {                                   // ABI v4: this pass's compositing and the next pass's depths from ONE launch
        UCNERF_REQUIRE(!p->u_sampled, "render_fused_fwd: resample and the per-sample uncertainty outputs exclude each other");
        s_res = *p->resample;
        s_res.weights = nullptr; s_res.z_merge = p->z;
    }
    if (p->cfg.precision == 3) {                        // row f1: gather + PE + MLP in one launch, no feature buffer at all
        UCNERF_REQUIRE(cl_given(p) && !keep_feats && !p->u_sampled && !p->train_workspace,
                       "render_fused_fwd: precision 3 (gather fused into the MLP kernel) needs the channel-last sources; it keeps no features "
                       "and returns no per-sample uncertainty");
        const bool gen = p->gen_rays != nullptr;          // rays, depths and direction features are generated inside the launch (w->angle: its per-sample scratch)
        const bool tail = tail_fits(p, p->resample ? &s_res : nullptr);
        // the view-direction features: given (dir_feat), generated with the rays (gen), made in the tail route's block prologues when the rotation
        // lives on the device (w2c_dir_dev: no launch of its own), or from ucnerf_dir_feature
        const bool dirs_in_tail = tail && !gen && !p->dir_feat && p->w2c_dir_dev;
        if (!gen && !p->dir_feat && !dirs_in_tail && (rc = launch_dirs(p, st, w))) return rc;
        raw_fused = p->raw ? p->raw : w->raw;
        if (p->ev_mlp_start && (rc = ucnerf_event_record(p->ev_mlp_start, st))) return rc;
        if (tail) composite_args(p, raw_fused, &c);
        // (rays generated inside the launch: the RAYGEN instantiation derives every lane's direction feature itself and uses w->angle as scratch; the
        //  tail route's blocks write the caller's buffers first and read them like given ones)
        if ((rc = launch_mlp_fwd_bf16x3_gather(p, gen && !tail ? w->angle : p->dir_feat ? p->dir_feat : w->angle, raw_fused, st, tail ? &c : nullptr,
                                               tail && p->resample ? &s_res : nullptr, dirs_in_tail ? w->angle : nullptr))) return rc;
        if (p->ev_mlp_stop && (rc = ucnerf_event_record(p->ev_mlp_stop, st))) return rc;
        if (tail) { ++g_tail_launches; return UCNERF_OK; }      // K7 (and K8, K9) ran inside the launch
    } else if (cl_given(p)) {                                  // fast path: channel-last sources, coordinates derived in-kernel
        g.out_tiled = keep_feats ? (p->feats_tiled ? 1 : 0) : 1;
        g.feats = keep_feats ? p->feats : w->feats;
        if ((rc = launch_gather_cl(p, g.feats, g.out_tiled, w->ndc, st))) return rc;
    } else {
        if (!coords_given(p) && (rc = launch_points(p, st, w))) return rc;
        gather_geometry(p, w, &g);
        g.out_tiled = keep_feats ? (p->feats_tiled ? 1 : 0) : 1;
        g.feats = keep_feats ? p->feats : w->feats;
        g.u_out = p->u_sampled;
        if ((rc = ucnerf_feat_gather_fwd(&g, st))) return rc;
    }
    ucnerf_mlp_params m;
    memset(&m, 0, sizeof(m));
    m.raw = raw_fused;
    if (!raw_fused) {
        if (!p->dir_feat && (rc = launch_dirs(p, st, w))) return rc;
        mlp_args(p, w, g.feats, g.out_tiled, p->raw ? p->raw : w->raw, &m);
        if (p->dir_feat) m.dirs = p->dir_feat;
        if (p->ev_mlp_start && (rc = ucnerf_event_record(p->ev_mlp_start, st))) return rc;
        if (p->train_workspace && keep_feats && p->raw) {    // training forward: activations go straight into the backward's workspace
            Workspace wb;
            float *g_raw, *g_feats, *mlp_ws;
            carve_bwd_render(p->train_workspace, p->n, p->S, V, &wb, &g_raw, &g_feats, &mlp_ws);
            if ((rc = ucnerf_mlp_fwd_train(&m, mlp_ws, p->train_bwd_mode, st))) return rc;
        } else if ((rc = ucnerf_mlp_fwd(&m, st))) return rc;
        if (p->ev_mlp_stop && (rc = ucnerf_event_record(p->ev_mlp_stop, st))) return rc;
    }

    composite_args(p, m.raw, &c);
    if (p->resample) return ucnerf_composite_sample_pdf(&c, &s_res, st);
    return ucnerf_composite_fwd(&c, st);
}

}  // namespace ucnerf

using namespace ucnerf;

extern "C" {

int64_t ucnerf_fused_tail_launches(void) { return (int64_t)g_tail_launches.load(); }
int32_t ucnerf_set_fused_tail(int32_t on) { tail_knobs_init(); return g_tail_on.exchange(on ? 1 : 0); }
int32_t ucnerf_fused_tail_fits(int32_t n, int32_t S) { return tail_size_fits(n, S) ? 1 : 0; }
int32_t ucnerf_fused_tail_fits_resample(int32_t n, int32_t S, int32_t n_samples) { return tail_size_fits(n, S) && tail_resample_fits(S, n_samples, 0) ? 1 : 0; }

int64_t ucnerf_render_workspace_floats(int32_t n, int32_t S, int32_t V) {
    if (n < 0 || S < 1 || V < 1 || V > 8) return fail(UCNERF_EINVAL, "render_workspace: bad sizes n=%d S=%d V=%d", n, S, V);
    Workspace w;
    return (int64_t)carve(nullptr, n, S, V, &w);
}

// backward workspace = [forward-style carve (points, coordinates, directions)] [g_raw 4M] [g_feats M*F] [mlp_bwd scratch]

int64_t ucnerf_render_bwd_workspace_floats(int32_t n, int32_t S, int32_t V) {
    if (n < 0 || S < 1 || V < 1 || V > 8) return fail(UCNERF_EINVAL, "render_bwd_workspace: bad sizes n=%d S=%d V=%d", n, S, V);
    Workspace w;
    float *a, *b, *c;
    return (int64_t)carve_bwd_render(nullptr, n, S, V, &w, &a, &b, &c);
}

int ucnerf_render_fused_bwd(const ucnerf_render_bwd_params* bp, void* stream) {
    UCNERF_REQUIRE(bp, "render_fused_bwd: null params");
    const ucnerf_render_params* p = &bp->fwd;
    UCNERF_COUNT(p->n);
    UCNERF_REQUIRE(p->rays_o && p->rays_d && p->z && p->wstream && p->raw && p->feats && bp->g_rgb && bp->flat_params && bp->g_flat &&
                       bp->workspace, "render_fused_bwd: null pointer (fwd.raw and fwd.feats must hold the forward's outputs)");
    UCNERF_REQUIRE(((uintptr_t)bp->workspace & 15) == 0, "render_fused_bwd: workspace must be 16-byte aligned");
    UCNERF_REQUIRE(!bp->saved_valid || p->train_bwd_mode == bp->bwd_mode, "render_fused_bwd: the forward kept its activations for bwd_mode %d, this call asks for %d "
                   "(the two modes read different formats)", p->train_bwd_mode, bp->bwd_mode);
    hipStream_t st = (hipStream_t)stream;
    const int V = p->cfg.n_src;
    Workspace w;
    float *g_raw, *g_feats, *mlp_ws;
    carve_bwd_render(bp->workspace, p->n, p->S, V, &w, &g_raw, &g_feats, &mlp_ws);
    int rc;
    ucnerf_render_params q = *p;            // points / coordinates / directions are recomputed into OUR workspace
    q.workspace = bp->workspace;
    if (!coords_given(&q) && (rc = launch_points(&q, st, &w))) return rc;
    if (!p->dir_feat && (rc = launch_dirs(&q, st, &w))) return rc;

    ucnerf_composite_bwd_params cb;
    memset(&cb, 0, sizeof(cb));
    cb.fwd.n = p->n; cb.fwd.S = p->S; cb.fwd.variant = 0; cb.fwd.white_bkgd = p->white_bkgd; cb.fwd.raw = p->raw; cb.fwd.z = p->z;
    cb.g_rgb = bp->g_rgb; cb.g_depth = bp->g_depth; cb.g_raw = g_raw;
    if ((rc = ucnerf_composite_bwd(&cb, st))) return rc;

    ucnerf_mlp_bwd_params mb;
    memset(&mb, 0, sizeof(mb));
    mlp_args(&q, &w, p->feats, p->feats_tiled ? 1 : 0, nullptr, &mb.fwd);
    if (p->dir_feat) mb.fwd.dirs = p->dir_feat;
    mb.fwd.raw = g_raw;                      // placeholder (not written by the backward)
    mb.g_raw = g_raw; mb.flat_params = bp->flat_params; mb.g_feats = g_feats; mb.g_flat = bp->g_flat; mb.workspace = mlp_ws;
    mb.saved_valid = bp->saved_valid; mb.bwd_mode = bp->bwd_mode;
    if ((rc = ucnerf_mlp_bwd(&mb, st))) return rc;

    if (bp->g_vol[0] || bp->g_vol[1] || bp->g_vol[2] || bp->g_conf || bp->g_img_feat || bp->g_cl.vol[0] || bp->g_cl.vol[1] || bp->g_cl.vol[2] || bp->g_cl.img_feat) {
        ucnerf_feat_gather_bwd_params gb;
        memset(&gb, 0, sizeof(gb));
        gather_geometry(&q, &w, &gb.fwd);
        gb.g_feats = g_feats;
        for (int k = 0; k < 3; ++k) gb.g_vol[k] = bp->g_vol[k];
        gb.g_conf = bp->g_conf; gb.g_img_feat = bp->g_img_feat; gb.scratch = bp->gather_scratch; gb.g_cl = bp->g_cl;
        if ((rc = ucnerf_feat_gather_bwd(&gb, st))) return rc;
    }
    return UCNERF_OK;
}

int ucnerf_render_fused_fwd(const ucnerf_render_params* p, void* stream) {
    UCNERF_REQUIRE(p, "render_fused_fwd: null params");
    UCNERF_COUNT(p->n);
    UCNERF_REQUIRE(p->rays_o && p->rays_d && p->z && p->workspace && p->wstream && p->rgb_map && p->depth_map,
                   "render_fused_fwd: null pointer");
    UCNERF_REQUIRE(p->S >= 1 && p->S <= 1024, "render_fused_fwd: S = %d outside 1..1024", p->S);
    UCNERF_REQUIRE((long long)p->n * p->S < (1ll << 31), "render_fused_fwd: n*S overflows int32");
    UCNERF_REQUIRE(((uintptr_t)p->workspace & 15) == 0, "render_fused_fwd: workspace must be 16-byte aligned");
    UCNERF_REQUIRE(!p->wu_map || p->u_sampled, "render_fused_fwd: wu_map needs u_sampled");
    const int n_given = !!p->pts_in + !!p->ndc1_in + !!p->ndc2_in + !!p->ndc3_in + !!p->ndc_in;
    UCNERF_REQUIRE(n_given == 0 || n_given == 5, "render_fused_fwd: pts_in / ndc1_in / ndc2_in / ndc3_in / ndc_in must be given together");
    UCNERF_REQUIRE((p->gen_rays != nullptr) == (p->gen_depths != nullptr), "render_fused_fwd: gen_rays and gen_depths go together");
    if (p->gen_rays) {
        const ucnerf_ray_gen_params* gr = p->gen_rays;
        const ucnerf_sample_stratified_params* gs = p->gen_depths;
        UCNERF_REQUIRE(p->cfg.precision == 3 && n_given == 0 && !p->near_far && !p->cl.bf16,
                       "render_fused_fwd: generated rays are served by the gather-fused kernel (cfg.precision 3) on derived coordinates, the scene's depth range and fp32 source copies");
        UCNERF_REQUIRE(gr->n == p->n && gs->n == p->n && gs->S == p->S, "render_fused_fwd: gen_rays / gen_depths sizes (%d, %d x %d) differ from the pass's (%d x %d)",
                       gr->n, gs->n, gs->S, p->n, p->S);
        UCNERF_REQUIRE(gr->xs && gr->ys && !gr->opengl && !gr->rays_o && !gr->pix, "render_fused_fwd: gen_rays takes pixel lists in the intrinsic-matrix convention (no origin / pixel outputs)");
        UCNERF_REQUIRE(!gs->rays && !gs->pts && (!(gs->perturb > 0.f) || gs->noise), "render_fused_fwd: gen_depths uses the scalar near / far form (and needs noise draws when perturb > 0)");
        UCNERF_REQUIRE(gs->near == p->near && gs->far == p->far, "render_fused_fwd: gen_depths samples [%g, %g] but the pass's scene range is [%g, %g]", gs->near, gs->far, p->near, p->far);
        UCNERF_REQUIRE(gr->rays_d && gr->angle && gs->z && gr->rays_d == p->rays_d && gs->z == p->z && gr->angle == p->dir_feat,
                       "render_fused_fwd: rays_d / z / dir_feat must be the buffers gen_rays / gen_depths fill");
    }
    Workspace w;
    return run_forward(p, (hipStream_t)stream, &w);
}

}  // extern "C"
