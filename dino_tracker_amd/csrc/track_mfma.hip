// placeholder until the fused path lands
#include "common.h"
size_t dtk_track_mfma_workspace_bytes(const dtk_geom*, int) { return 0; }
int dtk_track_mfma(const dtk_geom*, const float*, const float*, const void*, const float*, const float*, const int32_t*,
                   const int32_t*, const int32_t*, float*, int, const int32_t*, int, void*, size_t, void*) {
    dtk_set_error("dtk_track(mfma): not built");
    return DTK_E_INVALID;
}
extern "C" size_t dtk_feat_f16_bytes(const dtk_geom*) { return 0; }
extern "C" int dtk_make_feat_f16(const dtk_geom*, const float*, const float*, void*, void*) {
    dtk_set_error("dtk_make_feat_f16: not built");
    return DTK_E_INVALID;
}
