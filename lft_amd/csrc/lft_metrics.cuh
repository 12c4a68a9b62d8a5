// lft_metrics.cuh -- per-view PSNR and Gaussian-window SSIM of SR mosaics on the GPU: the numbers the reference logs
// through scikit-image (utils/utils.py:56-88 cal_metrics; oracle/metrics_oracle.py restates the algorithm).
// Mosaics are [B,1,A*h,A*w] fp32, view (u,v) = rows u*h.., cols v*w.. .  All sums in fp64 as scikit-image does.
#pragma once
#include <hip/hip_runtime.h>

constexpr int kMetTile = 16, kMetR = 5, kMetIn = kMetTile + 2 * kMetR;       // 16x16 outputs need a 26x26 input tile

// part[(view*ntiles + tile)*3 + {0,1,2}] = sum of the SSIM map over the tile's interior pixels, sum of squared error
// over the tile's pixels, min of the label over the tile's pixels.
__global__ __launch_bounds__(256) void k_view_metrics(const float* __restrict__ label, const float* __restrict__ out,
                                                      double* __restrict__ part, int A, int h, int w, double C1, double C2) {
    __shared__ float tx[kMetIn][kMetIn + 1], ty[kMetIn][kMetIn + 1];
    __shared__ double hz[5][kMetIn][kMetTile];
    __shared__ double red[3][256];
    const int tiles_x = (w + kMetTile - 1) / kMetTile, tiles_y = (h + kMetTile - 1) / kMetTile;
    const int tile = blockIdx.x, view = blockIdx.y;                                 // view = (b*A + u)*A + v
    const int ty0 = (tile / tiles_x) * kMetTile, tx0 = (tile % tiles_x) * kMetTile;
    const int v = view % A, u = (view / A) % A, b = view / (A * A);
    const size_t base = ((size_t)b * A * h + (size_t)u * h) * (A * w) + (size_t)v * w;
    double g[11];
    {
        double s = 0.0;
        for (int k = -kMetR; k <= kMetR; ++k) { g[k + kMetR] = exp(-0.5 * k * k / (1.5 * 1.5)); s += g[k + kMetR]; }
        for (int k = 0; k < 11; ++k) g[k] /= s;
    }
    for (int i = threadIdx.x; i < kMetIn * kMetIn; i += 256) {
        const int yy = ty0 - kMetR + i / kMetIn, xx = tx0 - kMetR + i % kMetIn;
        const bool in = yy >= 0 && yy < h && xx >= 0 && xx < w;                      // outside values are never used by an interior pixel
        tx[i / kMetIn][i % kMetIn] = in ? label[base + (size_t)yy * (A * w) + xx] : 0.0f;
        ty[i / kMetIn][i % kMetIn] = in ? out[base + (size_t)yy * (A * w) + xx] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kMetIn * kMetTile; i += 256) {                    // horizontal pass of the 5 maps
        const int rr = i / kMetTile, cc = i % kMetTile;
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int k = 0; k < 11; ++k) {
            const double a = tx[rr][cc + k], c = ty[rr][cc + k];
            sx += g[k] * a; sy += g[k] * c; sxx += g[k] * a * a; syy += g[k] * c * c; sxy += g[k] * a * c;
        }
        hz[0][rr][cc] = sx; hz[1][rr][cc] = sy; hz[2][rr][cc] = sxx; hz[3][rr][cc] = syy; hz[4][rr][cc] = sxy;
    }
    __syncthreads();
    const int ly = threadIdx.x / kMetTile, lx = threadIdx.x % kMetTile, y = ty0 + ly, x = tx0 + lx;
    double ssim = 0.0, se = 0.0, tmin = 1e300;
    if (y < h && x < w) {
        const double a = tx[ly + kMetR][lx + kMetR], c = ty[ly + kMetR][lx + kMetR];
        se = (a - c) * (a - c);
        tmin = a;
        if (y >= kMetR && y < h - kMetR && x >= kMetR && x < w - kMetR) {
            double m[5] = {0, 0, 0, 0, 0};
            for (int k = 0; k < 11; ++k)
                for (int q = 0; q < 5; ++q) m[q] += g[k] * hz[q][ly + k][lx];
            const double cov = 121.0 / 120.0;
            const double vx = cov * (m[2] - m[0] * m[0]), vy = cov * (m[3] - m[1] * m[1]), vxy = cov * (m[4] - m[0] * m[1]);
            ssim = ((2 * m[0] * m[1] + C1) * (2 * vxy + C2)) / ((m[0] * m[0] + m[1] * m[1] + C1) * (vx + vy + C2));
        }
    }
    red[0][threadIdx.x] = ssim; red[1][threadIdx.x] = se; red[2][threadIdx.x] = tmin;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) {
            red[0][threadIdx.x] += red[0][threadIdx.x + k];
            red[1][threadIdx.x] += red[1][threadIdx.x + k];
            red[2][threadIdx.x] = fmin(red[2][threadIdx.x], red[2][threadIdx.x + k]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* p = part + ((size_t)view * (tiles_x * tiles_y) + tile) * 3;
        p[0] = red[0][0]; p[1] = red[1][0]; p[2] = red[2][0];
    }
    (void)tiles_y;
}
__global__ void k_view_metrics_final(const double* __restrict__ part, int nviews, int ntiles, int h, int w,
                                     float* __restrict__ psnr, float* __restrict__ ssim) {
    const int view = blockIdx.x * blockDim.x + threadIdx.x;
    if (view >= nviews) return;
    double s = 0.0, se = 0.0, tmin = 1e300;
    for (int t = 0; t < ntiles; ++t) {
        const double* p = part + ((size_t)view * ntiles + t) * 3;
        s += p[0]; se += p[1]; tmin = fmin(tmin, p[2]);
    }
    const double range = tmin >= 0.0 ? 1.0 : 2.0;                                   // skimage: float images, dtype range [-1, 1]
    psnr[view] = (float)(10.0 * log10(range * range / (se / ((double)h * w))));
    ssim[view] = (float)(s / ((double)(h - 2 * kMetR) * (w - 2 * kMetR)));
}
