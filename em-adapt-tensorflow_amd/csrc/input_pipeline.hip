// Batched input pipeline (the reference's dataset.image_preprocess, dataset.py:147-185): everything behind
// the host's JPEG / PNG decode for a WHOLE BATCH in one launch, straight from the decoded bytes.
//
// Round 5 ran the reference's op list sample by sample -- uint8 -> f32, resize_bilinear to the random scale, resize_bilinear to the
// crop, resize_nearest twice for the label, BGR - mean + flip: nine launches, three synchronous host-to-device copies and ~40 trips
// through the interpreter lock per SAMPLE; measured in round 6 (profiles/r06_loader.txt): 150 images/s, below even the fp32 step.
// Here the host hands over one staging buffer holding the decoded samples as they come out of the decoder (uint8 RGB rows, uint8
// class indices) plus one descriptor per sample, and one thread per OUTPUT pixel walks the op list backwards:
//   out[y][x]            = BGR(resized2[y][flip ? OW - 1 - x : x]) - mean                      dataset.py:175-181,187-199
//   resized2[y][x]       = bilinear sample of resized1 at (y * h1 / OH, x * w1 / OW)            dataset.py:169-173
//   resized1[yy][xx]     = bilinear sample of the decoded image at (yy * h0 / h1, xx * w0 / w1) dataset.py:153-163 (train only)
// with TF 1.x's legacy source mapping (align_corners = False, no half-pixel offset) and, for the label, the nearest rule
// min(floor(dst * in / out), in - 1) applied twice.  A stage-2 tap needs four stage-1 values, each a blend of four decoded pixels:
// sixteen byte triples per output pixel, all L2-resident (a 500 x 375 sample is 0.56 MB).  The arithmetic is the two-kernel
// path's, operation for operation, under `fp contract(off)` (every multiply and add rounds on its own, like the restatement and
// TF's CPU kernels), so the batch comes out bit for bit as emadapt_pred_bilinear_argmax x 2 + emadapt_resize_nearest x 2 +
// emadapt_bgr_mean_flip did (tests/test_gpu_next_rows.py).
#include "common.h"

namespace emadapt {
namespace {

constexpr int kVocMaxBatch = 64;   // descriptors travel as kernel arguments: 64 x 40 bytes

struct VocBatch {
    emadapt_voc_sample s[kVocMaxBatch];
};

__device__ __forceinline__ float voc_lerp3(const unsigned char *__restrict__ img, int w0, int y0, int y1, int x0, int x1, float ly, float lx, int c) {
#pragma clang fp contract(off)
    const float tl = (float)img[((long long)y0 * w0 + x0) * 3 + c], tr = (float)img[((long long)y0 * w0 + x1) * 3 + c];
    const float bl = (float)img[((long long)y1 * w0 + x0) * 3 + c], br = (float)img[((long long)y1 * w0 + x1) * 3 + c];
    const float top = tl + (tr - tl) * lx;
    const float bot = bl + (br - bl) * lx;
    return top + (bot - top) * ly;
}

// one stage-1 pixel (yy, xx) of the randomly scaled image, channel c
__device__ __forceinline__ float voc_stage1(const unsigned char *__restrict__ img, int h0, int w0, float hs1, float ws1, int yy, int xx, int c) {
#pragma clang fp contract(off)
    const float fy = (float)yy * hs1, fx = (float)xx * ws1;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = y0 + 1 < h0 ? y0 + 1 : h0 - 1, x1 = x0 + 1 < w0 ? x0 + 1 : w0 - 1;
    return voc_lerp3(img, w0, y0, y1, x0, x1, fy - (float)y0, fx - (float)x0, c);
}

__global__ void __launch_bounds__(256) voc_preprocess_kernel(const unsigned char *__restrict__ staging, VocBatch batch, int B, int OH, int OW,
                                                           float m0, float m1, float m2, float *__restrict__ img_out,
                                                           float *__restrict__ lab_out) {
#pragma clang fp contract(off)
    const long long n = (long long)B * OH * OW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % OW);
        const int y = (int)((idx / OW) % OH);
        const int b = (int)(idx / ((long long)OW * OH));
        const emadapt_voc_sample &s = batch.s[b];
        const unsigned char *img = staging + s.img_off;
        const unsigned char *lab = staging + s.lab_off;
        const int xs = s.flip ? OW - 1 - x : x;
        const bool two = s.h1 > 0;   // the random-scale stage in front (the "train" branch, dataset.py:151-163)
        const int ih = two ? s.h1 : s.h0, iw = two ? s.w1 : s.w0;
        const float hs2 = (float)ih / (float)OH, ws2 = (float)iw / (float)OW;
        const float hs1 = two ? (float)s.h0 / (float)s.h1 : 1.f, ws1 = two ? (float)s.w0 / (float)s.w1 : 1.f;
        // ---- image: stage 2 (to OH x OW) over stage 1 (to h1 x w1) over the decoded bytes
        const float fy = (float)y * hs2, fx = (float)xs * ws2;
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        const int y1 = y0 + 1 < ih ? y0 + 1 : ih - 1, x1 = x0 + 1 < iw ? x0 + 1 : iw - 1;
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (two) {
                const float tl = voc_stage1(img, s.h0, s.w0, hs1, ws1, y0, x0, c), tr = voc_stage1(img, s.h0, s.w0, hs1, ws1, y0, x1, c);
                const float bl = voc_stage1(img, s.h0, s.w0, hs1, ws1, y1, x0, c), br = voc_stage1(img, s.h0, s.w0, hs1, ws1, y1, x1, c);
                const float top = tl + (tr - tl) * lx;
                const float bot = bl + (br - bl) * lx;
                rgb[c] = top + (bot - top) * ly;
            } else {
                rgb[c] = voc_lerp3(img, s.w0, y0, y1, x0, x1, ly, lx, c);
            }
        }
        img_out[idx * 3 + 0] = rgb[2] - m0;   // B
        img_out[idx * 3 + 1] = rgb[1] - m1;   // G
        img_out[idx * 3 + 2] = rgb[0] - m2;   // R
        // ---- label: nearest, twice
        if (lab_out) {
            int sy = (int)floorf((float)y * hs2), sx = (int)floorf((float)xs * ws2);
            sy = sy < ih - 1 ? sy : ih - 1;
            sx = sx < iw - 1 ? sx : iw - 1;
            if (two) {
                int ty = (int)floorf((float)sy * hs1), tx = (int)floorf((float)sx * ws1);
                sy = ty < s.h0 - 1 ? ty : s.h0 - 1;
                sx = tx < s.w0 - 1 ? tx : s.w0 - 1;
            }
            lab_out[idx] = (float)lab[(long long)sy * s.w0 + sx];
        }
    }
}

}  // namespace
}  // namespace emadapt

using namespace emadapt;

extern "C" {

int emadapt_voc_preprocess(const uint8_t *staging, int64_t staging_bytes, const emadapt_voc_sample *samples, int B, int OH, int OW,
                           const float mean_bgr[3], float *img_out, float *label_out, void *stream_) {
    if (!staging || !samples || !mean_bgr || !img_out || B < 0 || OH <= 0 || OW <= 0 || staging_bytes < 0) return EMADAPT_ERR_INVALID;
    if (B == 0) return EMADAPT_OK;
    for (int b = 0; b < B; b++) {   // host array: every offset and size is checked before anything is launched
        const emadapt_voc_sample &s = samples[b];
        if (s.h0 <= 0 || s.w0 <= 0 || s.h1 < 0 || s.w1 < 0 || (s.h1 > 0) != (s.w1 > 0)) return EMADAPT_ERR_INVALID;
        const int64_t px = (int64_t)s.h0 * s.w0;
        if (s.img_off < 0 || s.img_off + px * 3 > staging_bytes) return EMADAPT_ERR_INVALID;
        if (label_out && (s.lab_off < 0 || s.lab_off + px > staging_bytes)) return EMADAPT_ERR_INVALID;
        // fp32 holds every coordinate exactly while the sizes stay below 2^24
        if (s.h0 >= (1 << 24) || s.w0 >= (1 << 24) || s.h1 >= (1 << 24) || s.w1 >= (1 << 24)) return EMADAPT_ERR_UNSUPPORTED;
    }
    hipStream_t stream = as_stream(stream_);
    for (int b0 = 0; b0 < B; b0 += kVocMaxBatch) {
        const int nb = B - b0 < kVocMaxBatch ? B - b0 : kVocMaxBatch;
        VocBatch batch;
        for (int b = 0; b < nb; b++) batch.s[b] = samples[b0 + b];
        for (int b = nb; b < kVocMaxBatch; b++) batch.s[b] = samples[b0];
        const long long n = (long long)nb * OH * OW;
        int grid = (int)ceil_div(n, 256);
        if (grid > 16384) grid = 16384;
        ProfScope prof(EMADAPT_K_OTHER, 0.0, (double)n * 16.0, stream);
        hipLaunchKernelGGL(voc_preprocess_kernel, dim3(grid), dim3(256), 0, stream, staging, batch, nb, OH, OW, mean_bgr[0], mean_bgr[1],
                           mean_bgr[2], img_out + (long long)b0 * OH * OW * 3, label_out ? label_out + (long long)b0 * OH * OW : nullptr);
        const int rc = launch_status();
        if (rc) return rc;
    }
    return EMADAPT_OK;
}

}  // extern "C"
