// track.hip -- dtk_track dispatcher (exact fp32 path / fused MFMA path).
#include "common.h"

size_t dtk_track_exact_workspace_bytes(const dtk_geom* g, int M);
int dtk_track_exact(const dtk_geom* g, const float* feat, const float* norms, const float* head, const float* emb,
                    const int32_t* src_row, const int32_t* tgt, const int32_t* out_idx, float* out_xy, int M,
                    const int32_t* dM, int normalized, void* workspace, size_t workspace_bytes, void* stream);
size_t dtk_track_mfma_workspace_bytes(const dtk_geom* g, int M, int round_sources);
int dtk_track_mfma(const dtk_geom* g, const float* feat, const float* norms, const void* feat_f16, const float* head,
                   const float* emb, const int32_t* src_row, const int32_t* tgt, const int32_t* out_idx, float* out_xy,
                   int M, const int32_t* dM, const dtk_track_opts* opts, dtk_track_stats* stats, void* workspace,
                   size_t workspace_bytes, void* stream);

static int check_track_geom(const dtk_geom* g) {
    DTK_REQUIRE(g != nullptr, "dtk_track: null geometry");
    DTK_REQUIRE(g->T > 0 && g->C > 0 && g->ph > 0 && g->pw > 0 && g->patch > 0 && g->stride > 0 && g->radius >= 0.f,
                "dtk_track: bad geometry");
    DTK_REQUIRE(g->ph == 1 + (g->video_h - g->patch) / g->stride && g->pw == 1 + (g->video_w - g->patch) / g->stride,
                "dtk_track: token grid %dx%d inconsistent with video %dx%d", g->ph, g->pw, g->video_h, g->video_w);
    return DTK_OK;
}

extern "C" size_t dtk_track_workspace_bytes(const dtk_geom* g, int M, const dtk_track_opts* opts) {
    if (!g || M <= 0 || !opts) return 0;
    // (the MFMA figure already contains a region for the exact path, which re-does inconclusive sources)
    if (opts->method == DTK_TRACK_MFMA) return dtk_track_mfma_workspace_bytes(g, M, opts->round_sources);
    return dtk_track_exact_workspace_bytes(g, M);
}

extern "C" int dtk_track(const dtk_geom* g, const float* feat, const float* norms, const void* feat_f16,
                         const float* head, const float* emb, const int32_t* src_row, const int32_t* tgt,
                         const int32_t* out_idx, float* out_xy, int M, const int32_t* dM, const dtk_track_opts* opts,
                         dtk_track_stats* stats, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_track_geom(g);
    if (rc) return rc;
    DTK_REQUIRE(feat && norms && head && emb && tgt && out_xy && workspace && opts, "dtk_track: null pointer");
    DTK_REQUIRE(M >= 0, "dtk_track: negative M");
    DTK_REQUIRE(opts->round_sources >= 0 && (opts->tier == DTK_TIER_AUTO || opts->tier == DTK_TIER_WHOLE_MAP),
                "dtk_track: bad options (round_sources %d, tier %d)", opts->round_sources, opts->tier);
    if (stats) *stats = dtk_track_stats{0, 0, 0, 0};
    if (M == 0) return DTK_OK;
    if (opts->method == DTK_TRACK_EXACT) {
        if (stats) stats->sources = stats->exact_tier = M;  // (an upper bound when dM is given: the count stays on the device)
        return dtk_track_exact(g, feat, norms, head, emb, src_row, tgt, out_idx, out_xy, M, dM, opts->normalized, workspace,
                               workspace_bytes, stream);
    }
    if (opts->method == DTK_TRACK_MFMA) {
        DTK_REQUIRE(feat_f16 != nullptr, "dtk_track(mfma): feat_f16 is null (call dtk_make_feat_f16)");
        return dtk_track_mfma(g, feat, norms, feat_f16, head, emb, src_row, tgt, out_idx, out_xy, M, dM, opts, stats,
                              workspace, workspace_bytes, stream);
    }
    dtk_set_error("dtk_track: unknown method %d", opts->method);
    return DTK_E_INVALID;
}

int dtk_argmax_exact(const dtk_geom* g, const float* feat, const float* norms, const float* emb, const int32_t* src_row,
                     const int32_t* tgt, const int32_t* out_idx, int32_t* arg_cell, float* arg_cos, int M, void* workspace,
                     size_t workspace_bytes, void* stream);
int dtk_argmax_mfma(const dtk_geom* g, const float* feat, const float* norms, const void* feat_f16, const float* emb,
                    const int32_t* src_row, const int32_t* tgt, int32_t* arg_cell, float* arg_cos, int M, void* workspace,
                    size_t workspace_bytes, void* stream);

extern "C" int dtk_argmax_cells(const dtk_geom* g, const float* feat, const float* norms, const void* feat_f16,
                                const float* emb, const int32_t* src_row, const int32_t* tgt, int32_t* arg_cell,
                                float* arg_cos, int M, int method, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = check_track_geom(g);
    if (rc) return rc;
    DTK_REQUIRE(feat && norms && emb && tgt && arg_cell && arg_cos && workspace, "dtk_argmax_cells: null pointer");
    DTK_REQUIRE(M >= 0, "dtk_argmax_cells: negative M");
    if (M == 0) return DTK_OK;
    if (method == DTK_TRACK_MFMA) {
        DTK_REQUIRE(feat_f16 != nullptr, "dtk_argmax_cells(mfma): feat_f16 is null (call dtk_make_feat_f16)");
        return dtk_argmax_mfma(g, feat, norms, feat_f16, emb, src_row, tgt, arg_cell, arg_cos, M, workspace, workspace_bytes,
                               stream);
    }
    if (method == DTK_TRACK_EXACT)
        return dtk_argmax_exact(g, feat, norms, emb, src_row, tgt, nullptr, arg_cell, arg_cos, M, workspace, workspace_bytes,
                                stream);
    dtk_set_error("dtk_argmax_cells: unknown method %d", method);
    return DTK_E_INVALID;
}
