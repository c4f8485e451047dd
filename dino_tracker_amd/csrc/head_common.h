// head_common.h -- the tail of TrackerHead.forward shared by the exact and the MFMA paths.
#pragma once
#include "common.h"

// Lane 0 of the wave that summed the disk: zero-mass fallback (tracker_head.py:86-94), centre of mass, RangeNormalizer.
__device__ __forceinline__ void dtk_softargmax_finish(const dtk_geom& g, float sq, float sqx, float sqy, float cnt, float sx,
                                                      float sy, int normalized, float* out2) {
    if (sq < 1e-8f) {  // q <- (q + 1/|mask|) * mask
        const float uni = 1.f / cnt;
        sqx = sqx + uni * sx;
        sqy = sqy + uni * sy;
        sq = sq + uni * cnt;
    }
    const float xh = sqx / sq, yh = sqy / sq;
    float vx = 2.f * (xh / (float)(g.video_w - 1)) - 1.f;  // RangeNormalizer.forward, dst=(-1,1)
    float vy = 2.f * (yh / (float)(g.video_h - 1)) - 1.f;
    if (!normalized) {  // RangeNormalizer.unnormalize, src=(-1,1)
        vx = ((vx + 1.f) / 2.f) * (float)(g.video_w - 1);
        vy = ((vy + 1.f) / 2.f) * (float)(g.video_h - 1);
    }
    out2[0] = vx;
    out2[1] = vy;
}

// Shared with the MFMA path: finish one source from its (exact fp32) refined logits inside the disk.
// Runs on ONE wave.  zfun(row, col) returns z at a cell; zmax / Z are the softmax statistics of the whole map.
// Returns through lane 0.  (tracker_head.py:68-98,112,121 + model_inference.py:52)
template <typename ZF>
__device__ __forceinline__ void dtk_disk_softargmax(const dtk_geom& g, int kstar, float zmax, float Z, ZF zfun,
                                                    int normalized, float* out2, float* sq_report = nullptr) {
    const int lane = threadIdx.x & 63;
    const int rs = kstar / g.pw, cs = kstar % g.pw;
    const float half = (float)(g.patch / 2);
    const float px = (float)(cs * g.stride) + half, py = (float)(rs * g.stride) + half;
    const int R = (int)(g.radius / (float)g.stride) + 1;
    const int side = 2 * R + 1;
    float sq = 0.f, sqx = 0.f, sqy = 0.f, cnt = 0.f, sx = 0.f, sy = 0.f;
    for (int i = lane; i < side * side; i += WAVE) {
        const int r = rs - R + i / side, c = cs - R + i % side;
        if (r < 0 || r >= g.ph || c < 0 || c >= g.pw) continue;
        const float x = (float)(c * g.stride) + half, y = (float)(r * g.stride) + half;
        const float dx = x - px, dy = y - py;
        if (sqrtf(dx * dx + dy * dy) <= g.radius) {
            const float q = expf(zfun(r, c) - zmax) / Z;
            sq += q; sqx += q * x; sqy += q * y;
            cnt += 1.f; sx += x; sy += y;
        }
    }
    sq = wave_sum(sq); sqx = wave_sum(sqx); sqy = wave_sum(sqy);
    cnt = wave_sum(cnt); sx = wave_sum(sx); sy = wave_sum(sy);
    if (sq_report) *sq_report = sq;  // same value on every lane: lets the caller detect an undecidable fallback test
    if (lane == 0) dtk_softargmax_finish(g, sq, sqx, sqy, cnt, sx, sy, normalized, out2);
}
