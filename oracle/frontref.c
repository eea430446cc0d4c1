/*
 * frontref.c -- CPU restatement of the raster front-end (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the checker for the HIP kernels: a plain C restatement of the
 * OpenCV 4.x arithmetic behind the cv2 calls the reference's diagram parsers
 * make before their contour logic.  It is never linked into, imported by or
 * called from the product path; only tests, the smoke check and the benchmark's
 * CPU-baseline leg load it.
 *
 * The arithmetic lives in a third-party dependency of the reference that is not
 * vendored there: PyPI `opencv-python` (version unpinned by the reference; a
 * 4.10-4.12 era wheel is inferred from its CPython 3.13 bytecode caches).  Each
 * function restates the published OpenCV 4.x algorithm for one cv2 call and
 * cites the reference call sites it stands in for.
 *
 * PARITY PINNING: the reference ships no tests.  One artefact pair produced by
 * real cv2 (NewVersion2/inverted.png + NewVersion2/gray_img.png, written by
 * NewVersion2/text_detectionv2.py:22-30) pins fr_bgr2gray and fr_bitwise_not.
 * Every other function here is "parity unpinned": checked against hand-derived
 * known answers and an independent numpy/scipy restatement, not against cv2.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off matters: Hough rho and the float Gaussian must round each
 * multiply and add separately.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define FR_API __attribute__((visibility("default")))

typedef uint8_t u8;

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline u8 sat_u8(int v) { return (u8)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* ------------------------------------------------------------------ */
/* cv2.cvtColor(img, COLOR_BGR2GRAY) -- umlClassdiagram.py:14,1121;
 * umlSequencediagram.py:17,605; umlActivitydiagram.py:26,375,425;
 * umlStateDiagram.py:14,360; lineDetection.py:6.
 * OpenCV 4.x 8-bit path: Q15 coefficients B=3735 G=19235 R=9798, descale with
 * rounding. */
FR_API void fr_bgr2gray(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h)
{
#pragma omp parallel for
    for (int y = 0; y < h; y++) {
        const u8* s = src + (size_t)y * sp;
        u8* d = dst + (size_t)y * dp;
        for (int x = 0; x < w; x++)
            d[x] = (u8)((s[3 * x] * 3735 + s[3 * x + 1] * 19235 + s[3 * x + 2] * 9798 + 16384) >> 15);
    }
}

/* cv2.cvtColor(gray, COLOR_GRAY2BGR): replicate the channel. */
FR_API void fr_gray2bgr(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            u8 v = src[(size_t)y * sp + x];
            u8* d = dst + (size_t)y * dp + 3 * x;
            d[0] = d[1] = d[2] = v;
        }
}

/* cv2.cvtColor(img, COLOR_BGR2RGB / COLOR_RGB2BGR): swap channels 0 and 2
 * (NewVersion2/text_detectionv2.py:17). */
FR_API void fr_swap_rb(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const u8* s = src + (size_t)y * sp + 3 * x;
            u8* d = dst + (size_t)y * dp + 3 * x;
            u8 b = s[0], g = s[1], r = s[2];
            d[0] = r; d[1] = g; d[2] = b;
        }
}

/* cv2.bitwise_not (NewVersion2/text_detectionv2.py:22) / cv2.bitwise_or
 * ("New Version/textDetection copy.py":94).  `rowbytes` = w * channels. */
FR_API void fr_bitwise_not(const u8* src, size_t sp, u8* dst, size_t dp, int rowbytes, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < rowbytes; x++)
            dst[(size_t)y * dp + x] = (u8)~src[(size_t)y * sp + x];
}

FR_API void fr_bitwise_or(const u8* a, size_t ap, const u8* b, size_t bp, u8* dst, size_t dp,
                          int rowbytes, int h)
{
    for (int y = 0; y < h; y++)
        for (int x = 0; x < rowbytes; x++)
            dst[(size_t)y * dp + x] = a[(size_t)y * ap + x] | b[(size_t)y * bp + x];
}

/* ------------------------------------------------------------------ */
/* 256-bin histogram + Otsu's threshold as OpenCV's getThreshVal_Otsu_8u
 * ("New Version/textDetection.py":23 uses THRESH_BINARY+THRESH_OTSU). */
FR_API void fr_histogram(const u8* src, size_t sp, int w, int h, int* hist /*256*/)
{
    memset(hist, 0, 256 * sizeof(int));
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            hist[src[(size_t)y * sp + x]]++;
}

FR_API int fr_otsu_from_hist(const int* hist, long long npix)
{
    double mu = 0, scale = 1.0 / (double)npix;
    for (int i = 0; i < 256; i++)
        mu += i * (double)hist[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0;
    int max_val = 0;
    for (int i = 0; i < 256; i++) {
        double p_i, q2, mu2, sigma;
        p_i = hist[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        q2 = 1. - q1;
        if (fmin(q1, q2) < FLT_EPSILON || fmax(q1, q2) > 1. - FLT_EPSILON)
            continue;
        mu1 = (mu1 + i * p_i) / q1;
        mu2 = (mu - q1 * mu1) / q2;
        sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) {
            max_sigma = sigma;
            max_val = i;
        }
    }
    return max_val;
}

/* cv2.threshold(gray, t, maxval, type) -- umlClassdiagram.py:19 (t=150),
 * umlSequencediagram.py:22 and umlActivitydiagram.py:34 (128),
 * umlStateDiagram.py:19 (100), shapeDetection.py:10 (127), testing.py:50 (240),
 * "New Version/imageToBlackandWhite.py":12 (THRESH_BINARY).
 * type: 0 = BINARY, 1 = BINARY_INV, optionally | 8 (OTSU).  Returns the
 * threshold used, as cv2 does. */
FR_API double fr_threshold(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h,
                           double thresh, double maxval, int type)
{
    int ithresh;
    double ret;
    if (type & 8) {
        int hist[256];
        fr_histogram(src, sp, w, h, hist);
        ithresh = fr_otsu_from_hist(hist, (long long)w * h);
        ret = ithresh;
    } else {
        ithresh = (int)floor(thresh);
        ret = ithresh;
    }
    int imaxval = sat_u8((int)lrint(maxval));
    int inv = (type & 7) == 1;
    u8 hi = (u8)(inv ? 0 : imaxval), lo = (u8)(inv ? imaxval : 0);
#pragma omp parallel for
    for (int y = 0; y < h; y++) {
        const u8* s = src + (size_t)y * sp;
        u8* d = dst + (size_t)y * dp;
        for (int x = 0; x < w; x++)
            d[x] = (int)s[x] > ithresh ? hi : lo;
    }
    return ret;
}

/* ------------------------------------------------------------------ */
/* cv2.erode / cv2.dilate with an all-ones kw x kh kernel, anchor (ax, ay),
 * border pixels ignored -- "New Version/classDiagram copy.py":316-317,
 * "New Version/detectArrow.py":10-11, newVersion3/detect_arrow.py:39,
 * NewVersion2/detectArrowv2.py:24 (2x2).  One pass: the window of output x is
 * input [x - ax, x - ax + kw - 1] clipped to the image.  is_dilate: 0 = min,
 * 1 = max. */
static void morph_rows(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h, int k, int a,
                       int is_dilate)
{
    /* row padded with the identity on both sides, then k shifted min/max sweeps */
    const int padl = a, padr = k - 1 - a;
#pragma omp parallel
    {
        u8* buf = (u8*)malloc((size_t)w + padl + padr + 1);
#pragma omp for
        for (int y = 0; y < h; y++) {
            const u8* s = src + (size_t)y * sp;
            u8* d = dst + (size_t)y * dp;
            memset(buf, is_dilate ? 0 : 255, (size_t)w + padl + padr);
            memcpy(buf + padl, s, (size_t)w);
            memcpy(d, buf, (size_t)w);
            for (int i = 1; i < k; i++) {
                const u8* q = buf + i;
                if (is_dilate)
                    for (int x = 0; x < w; x++) d[x] = d[x] > q[x] ? d[x] : q[x];
                else
                    for (int x = 0; x < w; x++) d[x] = d[x] < q[x] ? d[x] : q[x];
            }
        }
        free(buf);
    }
}

static void morph_cols(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h, int k, int a,
                       int is_dilate)
{
#pragma omp parallel for
    for (int y = 0; y < h; y++) {
        u8* d = dst + (size_t)y * dp;
        int lo = imax(0, y - a), hi = imin(h - 1, y - a + k - 1);
        memset(d, is_dilate ? 0 : 255, (size_t)w);
        for (int i = lo; i <= hi; i++) {
            const u8* s = src + (size_t)i * sp;
            if (is_dilate)
                for (int x = 0; x < w; x++) d[x] = d[x] > s[x] ? d[x] : s[x];
            else
                for (int x = 0; x < w; x++) d[x] = d[x] < s[x] ? d[x] : s[x];
        }
    }
}

/* One erode (is_dilate=0) or dilate (1), `iterations` times, written the way
 * OpenCV evaluates an all-ones kernel: iterations fold into ONE pass with
 * kernel k + (n-1)(k-1) and anchor a*n (identical to iterating because border
 * samples are ignored).  iterations == 0 or a 1x1 kernel copies. */
FR_API void fr_morph_rect(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h, int kw, int kh,
                          int ax, int ay, int iterations, int is_dilate)
{
    if (ax < 0) ax = kw / 2;
    if (ay < 0) ay = kh / 2;
    if (iterations == 0 || (kw == 1 && kh == 1)) {
        for (int y = 0; y < h; y++)
            memcpy(dst + (size_t)y * dp, src + (size_t)y * sp, (size_t)w);
        return;
    }
    int n = iterations;
    int KW = kw + (n - 1) * (kw - 1), KH = kh + (n - 1) * (kh - 1);
    int AX = ax * n, AY = ay * n;
    u8* tmp = (u8*)malloc((size_t)w * h);
    morph_rows(src, sp, tmp, (size_t)w, w, h, KW, AX, is_dilate);
    morph_cols(tmp, (size_t)w, dst, dp, w, h, KH, AY, is_dilate);
    free(tmp);
}

/* cv2.morphologyEx(src, op, kernel, iterations=n) -- umlClassdiagram.py:27-28,
 * umlSequencediagram.py:30-31, umlActivitydiagram.py:41-42,
 * umlStateDiagram.py:27-28 (MORPH_OPEN, (1,15)/(15,1), n=2);
 * NewVersion2/img2UML.py:34-35 and newVersion3/detect_arrow.py:24,28 (CLOSE /
 * OPEN 3x3).  op: 0 erode, 1 dilate, 2 open (erode^n then dilate^n), 3 close. */
FR_API void fr_morphology_ex(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h, int op,
                             int kw, int kh, int ax, int ay, int iterations)
{
    if (op == 0 || op == 1) {
        fr_morph_rect(src, sp, dst, dp, w, h, kw, kh, ax, ay, iterations, op);
        return;
    }
    u8* tmp = (u8*)malloc((size_t)w * h);
    int first = (op == 2) ? 0 : 1;
    fr_morph_rect(src, sp, tmp, (size_t)w, w, h, kw, kh, ax, ay, iterations, first);
    fr_morph_rect(tmp, (size_t)w, dst, dp, w, h, kw, kh, ax, ay, iterations, !first);
    free(tmp);
}

/* ------------------------------------------------------------------ */
/* Border index maps (cv::borderInterpolate).  type: 0 CONSTANT (returns -1),
 * 1 REPLICATE, 2 REFLECT, 4 REFLECT_101. */
static int border_idx(int p, int len, int type)
{
    if ((unsigned)p < (unsigned)len) return p;
    if (type == 1) return p < 0 ? 0 : len - 1;
    if (type == 2 || type == 4) {
        int delta = (type == 4);
        if (len == 1) return 0;
        do {
            if (p < 0) p = -p - 1 + delta;
            else p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    return -1;
}

/* Gaussian taps in double (cv::getGaussianKernel): fixed tables for
 * sigma <= 0 and odd n <= 7, else exp(-x^2 / 2 sigma^2) normalised. */
static void gaussian_kernel_f64(int n, double sigma, double* k)
{
    static const double tab1[] = {1.0};
    static const double tab3[] = {0.25, 0.5, 0.25};
    static const double tab5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625};
    static const double tab7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    const double* fixed = NULL;
    if ((n & 1) && n <= 7 && sigma <= 0)
        fixed = n == 1 ? tab1 : n == 3 ? tab3 : n == 5 ? tab5 : tab7;
    if (fixed) {
        for (int i = 0; i < n; i++) k[i] = fixed[i];
        return;
    }
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        k[i] = exp(scale2X * x * x);
        sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] *= sum;
}

/* Q0.8 taps for the 8-bit fixed-point blur: exact tables for sigma <= 0 and
 * n in {1,3,5,7,9}; otherwise the double taps are quantised outside-in with
 * the rounding error carried to the next tap and the centre tap taking the
 * remainder so the taps sum to 256. */
FR_API void fr_gaussian_kernel_q8(int n, double sigma, int* k)
{
    if (sigma <= 0 && n == 9) {
        static const int t9[] = {4, 13, 30, 51, 60, 51, 30, 13, 4};
        memcpy(k, t9, sizeof(t9));
        return;
    }
    double* kd = (double*)malloc(sizeof(double) * (size_t)n);
    gaussian_kernel_f64(n, sigma, kd);
    if (sigma <= 0 && (n & 1) && n <= 7) {
        for (int i = 0; i < n; i++) k[i] = (int)lrint(kd[i] * 256.0);
    } else {
        int n2 = n / 2;
        double err = 0;
        long long sum = 0;
        for (int i = 0; i < n2; i++) {
            double adj = kd[i] * 256.0 + err;
            long long v = llrint(adj);
            err = adj - (double)v;
            k[i] = k[n - 1 - i] = (int)v;
            sum += v;
        }
        k[n2] = (int)(256 - 2 * sum);
    }
    free(kd);
}

/* cv2.medianBlur(gray, ksize) on 8-bit single-channel images, ksize 3 or 5 --
 * "New Version/textDetection.py":16 (3, feeding the Otsu threshold of :23),
 * "New Version/detectCircle.py":14 and "New Version/detectShape.py":11 (5).
 * OpenCV (modules/imgproc/src/median_blur.simd.hpp, medianBlur_SortNet) takes
 * the median of the ksize x ksize window with the rows and columns outside the
 * image replaced by the nearest one inside (BORDER_REPLICATE); a median is a
 * value of the window, so every correct implementation gives the same bytes. */
FR_API int fr_median_blur_u8(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h, int ksize)
{
    if ((ksize != 3 && ksize != 5) || w < 0 || h < 0) return -1;
    const int r = ksize / 2, n = ksize * ksize;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            u8 v[25];
            int m = 0;
            for (int dy = -r; dy <= r; dy++) {
                int yy = y + dy;
                yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
                for (int dx = -r; dx <= r; dx++) {
                    int xx = x + dx;
                    xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
                    v[m++] = src[(size_t)yy * sp + xx];
                }
            }
            /* insertion sort of at most 25 values */
            for (int i = 1; i < n; i++) {
                u8 t = v[i];
                int j = i - 1;
                while (j >= 0 && v[j] > t) {
                    v[j + 1] = v[j];
                    j--;
                }
                v[j + 1] = t;
            }
            dst[(size_t)y * dp + x] = v[n / 2];
        }
    }
    return 0;
}

/* cv2.GaussianBlur(gray, (kw,kh), sigmaX[, sigmaY, borderType]) on 8-bit
 * single-channel images -- NewVersion2/img2UML.py:26,
 * newVersion3/detect_arrow.py:16, newVersion3/detectHorizontalVerticalLines.py:58,
 * "Version4 with Chatgpt/contour_detection.py":13, "images/with API/withAPI.py":50
 * ((5,5),0); "New Version/detectArrow.py":7 ((5,5),1).
 * Fixed-point path: Q0.8 taps, horizontal pass exact in Q8.8, vertical pass in
 * Q16.16, output (acc + 32768) >> 16.  kw/kh <= 0 derives the size from sigma. */
FR_API int fr_gaussian_blur_u8(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h, int kw,
                               int kh, double sigmaX, double sigmaY, int border)
{
    if (sigmaY <= 0) sigmaY = sigmaX;
    if (kw <= 0 && sigmaX > 0) kw = (int)lrint(sigmaX * 3 * 2 + 1) | 1;
    if (kh <= 0 && sigmaY > 0) kh = (int)lrint(sigmaY * 3 * 2 + 1) | 1;
    if (kw <= 0 || kh <= 0 || !(kw & 1) || !(kh & 1)) return -1;
    sigmaX = fmax(sigmaX, 0.);
    sigmaY = fmax(sigmaY, 0.);
    int* kx = (int*)malloc(sizeof(int) * (size_t)kw);
    int* ky = (int*)malloc(sizeof(int) * (size_t)kh);
    fr_gaussian_kernel_q8(kw, sigmaX, kx);
    fr_gaussian_kernel_q8(kh, sigmaY, ky);
    int rx = kw / 2, ry = kh / 2;
    uint16_t* tmp = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)w * h);
#pragma omp parallel for
    for (int y = 0; y < h; y++) {
        const u8* s = src + (size_t)y * sp;
        for (int x = 0; x < w; x++) {
            uint32_t acc = 0;
            for (int i = 0; i < kw; i++) {
                int xi = border_idx(x + i - rx, w, border);
                uint32_t p = xi < 0 ? 0u : s[xi];
                acc += (uint32_t)kx[i] * p;
            }
            tmp[(size_t)y * w + x] = (uint16_t)(acc > 65535u ? 65535u : acc);
        }
    }
#pragma omp parallel for
    for (int y = 0; y < h; y++) {
        u8* d = dst + (size_t)y * dp;
        for (int x = 0; x < w; x++) {
            uint64_t acc = 0;
            for (int i = 0; i < kh; i++) {
                int yi = border_idx(y + i - ry, h, border);
                uint32_t p = yi < 0 ? 0u : tmp[(size_t)yi * w + x];
                acc += (uint64_t)(uint32_t)ky[i] * p;
            }
            uint64_t v = (acc + 32768u) >> 16;
            d[x] = (u8)(v > 255 ? 255 : v);
        }
    }
    free(tmp);
    free(kx);
    free(ky);
    return 0;
}

/* cv2.adaptiveThreshold(src, maxValue, method, type, blockSize, C) --
 * NewVersion2/img2UML.py:29 (GAUSSIAN_C, BINARY, 11, 2),
 * newVersion3/detect_arrow.py:19 (GAUSSIAN_C, BINARY_INV, 11, 2),
 * "Version4 with Chatgpt/textDetection.py":15 (GAUSSIAN_C, BINARY, 31, 2).
 * method: 0 MEAN_C (normalised box filter on 8-bit), 1 GAUSSIAN_C (float32
 * Gaussian, row pass accumulating taps left to right, column pass
 * k0*x0 + sum k_i*(x_+i + x_-i), each multiply and add rounded separately,
 * result rounded half-to-even to 8 bits).  Border REPLICATE for both.
 * `mean_out` (optional) receives the 8-bit local mean. */
FR_API int fr_adaptive_threshold(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h,
                                 double maxValue, int method, int type, int blockSize, double C,
                                 u8* mean_out)
{
    if (blockSize % 2 != 1 || blockSize <= 1) return -1;
    if (type != 0 && type != 1) return -1;
    if (method != 0 && method != 1) return -1;
    if (maxValue < 0) {
        for (int y = 0; y < h; y++) memset(dst + (size_t)y * dp, 0, (size_t)w);
        return 0;
    }
    int b = blockSize, r = b / 2;
    u8* mean = (u8*)malloc((size_t)w * h);
    if (method == 1) {
        double* kd = (double*)malloc(sizeof(double) * (size_t)b);
        float* kf = (float*)malloc(sizeof(float) * (size_t)b);
        gaussian_kernel_f64(b, 0.0, kd);
        for (int i = 0; i < b; i++) kf[i] = (float)kd[i];
        float* tmp = (float*)malloc(sizeof(float) * (size_t)w * h);
#pragma omp parallel for
        for (int y = 0; y < h; y++) {
            const u8* s = src + (size_t)y * sp;
            for (int x = 0; x < w; x++) {
                float acc = 0.f;
                for (int i = 0; i < b; i++) {
                    float p = (float)s[border_idx(x + i - r, w, 1)];
                    float prod = p * kf[i];
                    acc = acc + prod;
                }
                tmp[(size_t)y * w + x] = acc;
            }
        }
#pragma omp parallel for
        for (int y = 0; y < h; y++) {
            for (int x = 0; x < w; x++) {
                float acc = kf[r] * tmp[(size_t)y * w + x];
                for (int i = 1; i <= r; i++) {
                    float a = tmp[(size_t)border_idx(y + i, h, 1) * w + x];
                    float c = tmp[(size_t)border_idx(y - i, h, 1) * w + x];
                    float sum = a + c;
                    float prod = kf[r + i] * sum;
                    acc = acc + prod;
                }
                long v = lrintf(acc);
                mean[(size_t)y * w + x] = sat_u8((int)v);
            }
        }
        free(tmp);
        free(kd);
        free(kf);
    } else {
        int* tmp = (int*)malloc(sizeof(int) * (size_t)w * h);
        for (int y = 0; y < h; y++) {
            const u8* s = src + (size_t)y * sp;
            for (int x = 0; x < w; x++) {
                int acc = 0;
                for (int i = -r; i <= r; i++) acc += s[border_idx(x + i, w, 1)];
                tmp[(size_t)y * w + x] = acc;
            }
        }
        double scale = 1.0 / ((double)b * b);
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int acc = 0;
                for (int i = -r; i <= r; i++) acc += tmp[(size_t)border_idx(y + i, h, 1) * w + x];
                mean[(size_t)y * w + x] = sat_u8((int)lrint(acc * scale));
            }
        free(tmp);
    }
    int imaxval = sat_u8((int)lrint(maxValue));
    int idelta = type == 0 ? (int)ceil(C) : (int)floor(C);
    u8 tab[768];
    for (int i = 0; i < 768; i++)
        tab[i] = type == 0 ? (u8)(i - 255 > -idelta ? imaxval : 0) : (u8)(i - 255 <= -idelta ? imaxval : 0);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            dst[(size_t)y * dp + x] = tab[(int)src[(size_t)y * sp + x] - mean[(size_t)y * w + x] + 255];
    if (mean_out) memcpy(mean_out, mean, (size_t)w * h);
    free(mean);
    return 0;
}

/* ------------------------------------------------------------------ */
/* cv2.Canny(gray, t1, t2, apertureSize=3, L2gradient=False) --
 * lineDetection.py:9, "New Version/comparison.py":13,
 * "Version4 with Chatgpt/contour_detection.py":61,
 * newVersion3/detectHorizontalVerticalLines.py:61, "New Version/detectArrow.py":8.
 * Sobel 3x3 (BORDER_REPLICATE) -> |dx|+|dy| -> non-maximum suppression in four
 * sectors with tan(22.5 deg) in Q15 -> hysteresis over 8-neighbours.
 * `map_out` (optional, w*h) receives the pre-hysteresis map: 2 strong,
 * 0 weak candidate, 1 suppressed. */
FR_API void fr_canny(const u8* src, size_t sp, u8* dst, size_t dp, int w, int h, double t1, double t2,
                     u8* map_out)
{
    if (t1 > t2) { double t = t1; t1 = t2; t2 = t; }
    int low = (int)floor(t1), high = (int)floor(t2);
    size_t n = (size_t)w * h;
    short* dx = (short*)malloc(sizeof(short) * n);
    short* dy = (short*)malloc(sizeof(short) * n);
    int* mag = (int*)malloc(sizeof(int) * n);
    u8* map = (u8*)malloc(n);
#pragma omp parallel for
    for (int y = 0; y < h; y++) {
        const u8* r0 = src + (size_t)imax(y - 1, 0) * sp;
        const u8* r1 = src + (size_t)y * sp;
        const u8* r2 = src + (size_t)imin(y + 1, h - 1) * sp;
        for (int x = 0; x < w; x++) {
            int xl = imax(x - 1, 0), xr = imin(x + 1, w - 1);
            int gx = (r0[xr] - r0[xl]) + 2 * (r1[xr] - r1[xl]) + (r2[xr] - r2[xl]);
            int gy = (r2[xl] - r0[xl]) + 2 * (r2[x] - r0[x]) + (r2[xr] - r0[xr]);
            dx[(size_t)y * w + x] = (short)gx;
            dy[(size_t)y * w + x] = (short)gy;
            mag[(size_t)y * w + x] = abs(gx) + abs(gy);
        }
    }
#define MAG(yy, xx) (((yy) < 0 || (yy) >= h || (xx) < 0 || (xx) >= w) ? 0 : mag[(size_t)(yy) * w + (xx)])
#pragma omp parallel for
    for (int y = 0; y < h; y++) {
        for (int x = 0; x < w; x++) {
            int m = mag[(size_t)y * w + x];
            u8 v = 1;
            if (m > low) {
                int xs = dx[(size_t)y * w + x], ys = dy[(size_t)y * w + x];
                int ax = abs(xs), ay = abs(ys) << 15;
                int tg22x = ax * 13573;
                int is_max;
                if (ay < tg22x) {
                    is_max = m > MAG(y, x - 1) && m >= MAG(y, x + 1);
                } else {
                    int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) {
                        is_max = m > MAG(y - 1, x) && m >= MAG(y + 1, x);
                    } else {
                        int s = ((xs ^ ys) < 0) ? -1 : 1;
                        is_max = m > MAG(y - 1, x - s) && m > MAG(y + 1, x + s);
                    }
                }
                if (is_max) v = (m > high) ? 2 : 0;
            }
            map[(size_t)y * w + x] = v;
        }
    }
#undef MAG
    if (map_out) memcpy(map_out, map, n);
    /* hysteresis: flood from every strong pixel through weak candidates */
    /* (pixel indices as 32-bit unsigned numbers: an image has fewer than 2^32 pixels, but may have more than 2^31) */
    uint32_t* stack = (uint32_t*)malloc(sizeof(uint32_t) * n);
    size_t top = 0;
    for (size_t i = 0; i < n; i++)
        if (map[i] == 2) stack[top++] = (uint32_t)i;
    while (top) {
        uint32_t i = stack[--top];
        int y = (int)(i / (uint32_t)w), x = (int)(i % (uint32_t)w);
        for (int ddy = -1; ddy <= 1; ddy++)
            for (int ddx = -1; ddx <= 1; ddx++) {
                int yy = y + ddy, xx = x + ddx;
                if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                size_t j = (size_t)yy * w + xx;
                if (map[j] == 0) {
                    map[j] = 2;
                    stack[top++] = (uint32_t)j;
                }
            }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            dst[(size_t)y * dp + x] = map[(size_t)y * w + x] == 2 ? 255 : 0;
    free(stack);
    free(map);
    free(mag);
    free(dy);
    free(dx);
}

/* ------------------------------------------------------------------ */
/* cv2.HoughLines(img, rho, theta, threshold) -- "New Version/comparison.py":20,30.
 * Geometry of the vote accumulator (cv::HoughLinesStandard, min_theta = 0,
 * max_theta = pi). */
FR_API void fr_hough_dims(int w, int h, double rho, double theta, int* numangle, int* numrho)
{
    float frho = (float)rho, ftheta = (float)theta;
    int max_rho = w + h, min_rho = -max_rho;
    double min_theta = 0, max_theta = M_PI;
    int na = (int)floor((max_theta - min_theta) / ftheta) + 1;
    if (na > 1 && fabs(M_PI - (na - 1) * ftheta) < ftheta / 2) --na;
    *numangle = na;
    *numrho = (int)lrint(((max_rho - min_rho) + 1) / frho);
}

/* sin/cos tables: the angle accumulates in float32, the products with 1/rho
 * are taken in double and rounded to float32 once. */
FR_API void fr_hough_trig(double rho, double theta, int numangle, float* tabSin, float* tabCos)
{
    float frho = (float)rho, ftheta = (float)theta;
    float irho = 1 / frho;
    float ang = 0.f;
    for (int n = 0; n < numangle; ang += ftheta, n++) {
        tabSin[n] = (float)(sin((double)ang) * irho);
        tabCos[n] = (float)(cos((double)ang) * irho);
    }
}

/* Vote accumulator int32[(numangle+2) * (numrho+2)], zeroed here. */
FR_API void fr_hough_accum(const u8* src, size_t sp, int w, int h, double rho, double theta,
                           int* accum, int numangle, int numrho)
{
    float* tabSin = (float*)malloc(sizeof(float) * (size_t)numangle);
    float* tabCos = (float*)malloc(sizeof(float) * (size_t)numangle);
    fr_hough_trig(rho, theta, numangle, tabSin, tabCos);
    memset(accum, 0, sizeof(int) * (size_t)(numangle + 2) * (numrho + 2));
    /* non-zero pixels first (row-major, as OpenCV visits them), then votes */
    size_t npts = 0;
    int* pts = (int*)malloc(sizeof(int) * 2 * (size_t)w * h);
    for (int i = 0; i < h; i++)
        for (int j = 0; j < w; j++)
            if (src[(size_t)i * sp + j] != 0) {
                pts[2 * npts] = i;
                pts[2 * npts + 1] = j;
                npts++;
            }
#pragma omp parallel for
    for (int n = 0; n < numangle; n++) {
        int* row = accum + (size_t)(n + 1) * (numrho + 2);
        for (size_t k = 0; k < npts; k++) {
            float a = (float)pts[2 * k + 1] * tabCos[n];
            float b = (float)pts[2 * k] * tabSin[n];
            float s = a + b;
            int r = (int)lrintf(s);
            r += (numrho - 1) / 2;
            row[r + 1]++;
        }
    }
    free(pts);
    free(tabSin);
    free(tabCos);
}

typedef struct { int idx; int cnt; } fr_peak;
static int peak_cmp(const void* a, const void* b)
{
    const fr_peak* p = (const fr_peak*)a;
    const fr_peak* q = (const fr_peak*)b;
    if (p->cnt != q->cnt) return p->cnt > q->cnt ? -1 : 1;
    return p->idx < q->idx ? -1 : (p->idx > q->idx ? 1 : 0);
}

/* Local maxima above `threshold`, ordered by (votes desc, index asc), emitted
 * as (rho, theta) float32 pairs.  Returns the number of lines found; at most
 * `max_lines` are written. */
FR_API int fr_hough_peaks(const int* accum, int numangle, int numrho, double rho, double theta,
                          int threshold, float* lines, int max_lines)
{
    float frho = (float)rho, ftheta = (float)theta;
    size_t cap = 1024, cnt = 0;
    fr_peak* buf = (fr_peak*)malloc(sizeof(fr_peak) * cap);
    for (int r = 0; r < numrho; r++)
        for (int n = 0; n < numangle; n++) {
            int base = (n + 1) * (numrho + 2) + r + 1;
            if (accum[base] > threshold && accum[base] > accum[base - 1] &&
                accum[base] >= accum[base + 1] && accum[base] > accum[base - numrho - 2] &&
                accum[base] >= accum[base + numrho + 2]) {
                if (cnt == cap) {
                    cap *= 2;
                    buf = (fr_peak*)realloc(buf, sizeof(fr_peak) * cap);
                }
                buf[cnt].idx = base;
                buf[cnt].cnt = accum[base];
                cnt++;
            }
        }
    qsort(buf, cnt, sizeof(fr_peak), peak_cmp);
    double scale = 1. / (numrho + 2);
    for (size_t i = 0; i < cnt && (int)i < max_lines; i++) {
        int idx = buf[i].idx;
        int n = (int)floor(idx * scale) - 1;
        int r = idx - (n + 1) * (numrho + 2) - 1;
        lines[2 * i] = (r - (numrho - 1) * 0.5f) * frho;
        lines[2 * i + 1] = 0.f + n * ftheta;
    }
    free(buf);
    return (int)cnt;
}

/* ---- external contours -> bounding rectangles ------------------------------------------
 * get_segments of the parsers (umlClassdiagram.py:33-36, umlSequencediagram.py:36-39,
 * umlActivitydiagram.py:47-50, umlStateDiagram.py:33-36):
 *     contours, _ = cv2.findContours(lines, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)
 *     segments = [cv2.boundingRect(c) for c in contours]
 * Restated from the published algorithm (Suzuki & Abe 1985, "Topological structural analysis
 * of digitized binary images by border following") as OpenCV's findContours applies it: the
 * image is binarised (non-zero -> 1) inside a one-pixel zero frame and scanned in raster order;
 * a pixel 1 whose left neighbour is 0 starts an outer border; in RETR_EXTERNAL mode the border
 * is followed only if the last border pixel met on the row (LNBD) does not carry a positive
 * mark, i.e. the start does not lie inside an already followed border; followed pixels are
 * marked +2, or -126 where the border has background on its right.  Contours come out in
 * reverse order of discovery (each new contour is linked in front of its siblings).  The
 * bounding rectangle of the followed border is the bounding rectangle of the component.
 * Returns the number of rectangles found; at most `cap` are written as (x, y, w, h). */
FR_API int fr_external_boxes(const u8* src, size_t sp, int w, int h, int* boxes, int cap)
{
    if (w <= 0 || h <= 0) return 0;
    const int step = w + 2;
    signed char* img = (signed char*)calloc((size_t)step * (h + 2), 1);
    int* found = NULL;
    int nfound = 0, room = 0;
    if (!img) return -1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * step + x + 1] = src[(size_t)y * sp + x] ? 1 : 0;
    /* E, NE, N, NW, W, SW, S, SE: counter-clockwise on the screen, twice */
    int deltas[16];
    const int d8[8] = {1, -step + 1, -step, -step - 1, -1, step - 1, step, step + 1};
    for (int i = 0; i < 16; i++) deltas[i] = d8[i & 7];
    for (int y = 1; y <= h; y++) {
        signed char* row = img + (size_t)y * step;
        int prev = 0, lnbd = 0; /* LNBD restarts at the frame column of every row */
        for (int x = 1; x <= w; x++) {
            int p = row[x];
            if (p == prev) continue;
            if (prev == 0 && p == 1 && row[lnbd] <= 0) {
                /* follow the outer border that starts here */
                signed char* i0 = row + x;
                signed char *i1, *i3, *i4 = NULL;
                int s = 4, s_end = 4;
                int x0 = x, x1 = x, y0 = y, y1 = y;
                do {
                    s = (s - 1) & 7;
                    i1 = i0 + deltas[s];
                } while (*i1 == 0 && s != s_end);
                if (s == s_end) {
                    *i0 = (signed char)(2 | -128); /* isolated pixel */
                } else {
                    i3 = i0;
                    for (;;) {
                        s_end = s;
                        while (s < 15) {
                            i4 = i3 + deltas[++s];
                            if (*i4 != 0) break;
                        }
                        s &= 7;
                        if ((unsigned)(s - 1) < (unsigned)s_end)
                            *i3 = (signed char)(2 | -128); /* background on the right of the border */
                        else if (*i3 == 1)
                            *i3 = 2;
                        {
                            const long off = (long)(i3 - img);
                            const int py = (int)(off / step), px = (int)(off % step);
                            if (px < x0) x0 = px;
                            if (px > x1) x1 = px;
                            if (py < y0) y0 = py;
                            if (py > y1) y1 = py;
                        }
                        if (i4 == i0 && i3 == i1) break;
                        i3 = i4;
                        s = (s + 4) & 7;
                    }
                }
                if (nfound == room) {
                    room = room ? room * 2 : 256;
                    int* nf = (int*)realloc(found, sizeof(int) * 4 * (size_t)room);
                    if (!nf) {
                        free(found);
                        free(img);
                        return -1;
                    }
                    found = nf;
                }
                found[4 * nfound + 0] = x0 - 1;
                found[4 * nfound + 1] = y0 - 1;
                found[4 * nfound + 2] = x1 - x0 + 1;
                found[4 * nfound + 3] = y1 - y0 + 1;
                nfound++;
                p = row[x];
            }
            prev = p;
            if (prev & -2) lnbd = x; /* a marked border pixel */
        }
    }
    for (int i = 0; i < nfound && i < cap; i++)
        for (int k = 0; k < 4; k++) boxes[4 * i + k] = found[4 * (nfound - 1 - i) + k];
    free(found);
    free(img);
    return nfound;
}


/* ------------------------------------------------------------------ */
/* cv2.HoughLinesP(edges, rho, theta, threshold, minLineLength=, maxLineGap=) --
 * lineDetection.py:12-19, "New Version/comparison.py":14, detectHorizontalVerticalLines.py:64
 * and three more scripts.  Restatement of OpenCV 4.x's progressive probabilistic Hough
 * transform (cv::HoughLinesProbabilistic, hough.cpp): the non-zero pixels are visited in a
 * pseudo-random order drawn from cv::RNG(0xffffffffffffffff) (multiply-with-carry,
 * coefficient 4164903690); every visited pixel that still belongs to no line votes for all
 * angles; when a bin reaches the threshold the line is walked in both directions in 16.16
 * fixed point, gaps of up to maxLineGap allowed; the pixels of the walked segment leave the
 * mask and, if the segment is long enough, take their votes back.
 * lines: int32[cap][4] = (x0, y0, x1, y1); returns the number of segments found. */
static unsigned frp_rng_next(unsigned long long* st)
{
    *st = (unsigned long long)(unsigned)(*st) * 4164903690ULL + (unsigned)(*st >> 32);
    return (unsigned)(*st);
}

FR_API void fr_hough_p_dims(int w, int h, double rho, double theta, int* numangle, int* numrho)
{
    float frho = (float)rho, ftheta = (float)theta;
    int na = (int)floor((M_PI - 0.0) / ftheta) + 1;
    if (na > 1 && fabs(M_PI - (na - 1) * ftheta) < ftheta / 2) --na;
    *numangle = na;
    *numrho = (int)lrintf(((w + h) * 2 + 1) / frho);
}

FR_API int fr_hough_lines_p(const u8* src, size_t sp, int w, int h, double rho_, double theta_, int threshold,
                            int line_length, int line_gap, int* lines, int cap)
{
    if (w <= 0 || h <= 0) return 0;
    const float rho = (float)rho_, theta = (float)theta_;
    const float irho = 1 / rho;
    int numangle, numrho;
    fr_hough_p_dims(w, h, rho_, theta_, &numangle, &numrho);
    int* accum = (int*)calloc((size_t)numangle * numrho, sizeof(int));
    u8* mask = (u8*)malloc((size_t)w * h);
    float* ttab = (float*)malloc(sizeof(float) * 2 * (size_t)numangle);
    int* nz = (int*)malloc(sizeof(int) * 2 * (size_t)w * h);
    if (!accum || !mask || !ttab || !nz) { free(accum); free(mask); free(ttab); free(nz); return -1; }
    for (int n = 0; n < numangle; n++) {
        ttab[2 * n] = (float)(cos((double)n * theta) * irho);
        ttab[2 * n + 1] = (float)(sin((double)n * theta) * irho);
    }
    int count = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int on = src[(size_t)y * sp + x] != 0;
            mask[(size_t)y * w + x] = (u8)on;
            if (on) { nz[2 * count] = x; nz[2 * count + 1] = y; count++; }
        }
    unsigned long long rng = 0xffffffffffffffffULL;
    int nlines = 0;
    const int shift = 16;
    for (; count > 0; count--) {
        const int idx = (int)(frp_rng_next(&rng) % (unsigned)count);
        const int j = nz[2 * idx], i = nz[2 * idx + 1];
        nz[2 * idx] = nz[2 * (count - 1)];
        nz[2 * idx + 1] = nz[2 * (count - 1) + 1];
        if (!mask[(size_t)i * w + j]) continue;
        int max_val = threshold - 1, max_n = 0;
        for (int n = 0; n < numangle; n++) {
            const float fr = (float)j * ttab[2 * n] + (float)i * ttab[2 * n + 1];
            const int r = (int)lrintf(fr) + (numrho - 1) / 2;
            const int val = ++accum[(size_t)n * numrho + r];
            if (max_val < val) { max_val = val; max_n = n; }
        }
        if (max_val < threshold) continue;
        const float a = -ttab[2 * max_n + 1], b = ttab[2 * max_n];
        int x0 = j, y0 = i, dx0, dy0, xflag;
        if (fabsf(a) > fabsf(b)) {
            xflag = 1;
            dx0 = a > 0 ? 1 : -1;
            dy0 = (int)lrintf(b * (float)(1 << shift) / fabsf(a));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = 0;
            dy0 = b > 0 ? 1 : -1;
            dx0 = (int)lrintf(a * (float)(1 << shift) / fabsf(b));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        int ex[2] = {0, 0}, ey[2] = {0, 0};
        for (int k = 0; k < 2; k++) {
            int gap = 0, x = x0, y = y0, dx = dx0, dy = dy0;
            if (k > 0) dx = -dx, dy = -dy;
            for (;; x += dx, y += dy) {
                const int j1 = xflag ? x : x >> shift, i1 = xflag ? y >> shift : y;
                if (j1 < 0 || j1 >= w || i1 < 0 || i1 >= h) break;
                if (mask[(size_t)i1 * w + j1]) {
                    gap = 0;
                    ey[k] = i1;
                    ex[k] = j1;
                } else if (++gap > line_gap)
                    break;
            }
        }
        const int good = abs(ex[1] - ex[0]) >= line_length || abs(ey[1] - ey[0]) >= line_length;
        for (int k = 0; k < 2; k++) {
            int x = x0, y = y0, dx = dx0, dy = dy0;
            if (k > 0) dx = -dx, dy = -dy;
            for (;; x += dx, y += dy) {
                const int j1 = xflag ? x : x >> shift, i1 = xflag ? y >> shift : y;
                u8* m = mask + (size_t)i1 * w + j1;
                if (*m) {
                    if (good)
                        for (int n = 0; n < numangle; n++) {
                            const float fr = (float)j1 * ttab[2 * n] + (float)i1 * ttab[2 * n + 1];
                            accum[(size_t)n * numrho + (int)lrintf(fr) + (numrho - 1) / 2]--;
                        }
                    *m = 0;
                }
                if (i1 == ey[k] && j1 == ex[k]) break;
            }
        }
        if (good) {
            if (nlines < cap) {
                lines[4 * nlines] = ex[0]; lines[4 * nlines + 1] = ey[0];
                lines[4 * nlines + 2] = ex[1]; lines[4 * nlines + 3] = ey[1];
            }
            nlines++;
        }
    }
    free(accum); free(mask); free(ttab); free(nz);
    return nlines;
}

/* ------------------------------------------------------------------ */
/* The caller compositions of the reference, restated for the CPU baseline:
 * load_image_from_file / threshold_image / extract_lines
 * (umlClassdiagram.py:12-30 and siblings) followed by the line-detection
 * prefix (lineDetection.py:6-9) and the standard Hough accumulator
 * ("New Version/comparison.py":30).  Planes are w*h contiguous; accum is
 * (numangle+2)*(numrho+2) int32. */
FR_API void fr_front_end(const u8* bgr, int w, int h, int t, int line_len, int iterations, double c_lo,
                         double c_hi, u8* gray, u8* thresh, u8* vlines, u8* hlines, u8* edges,
                         int* accum, int numangle, int numrho)
{
    fr_bgr2gray(bgr, (size_t)3 * w, gray, (size_t)w, w, h);
    fr_threshold(gray, (size_t)w, thresh, (size_t)w, w, h, t, 255, 1);
    fr_morphology_ex(thresh, (size_t)w, vlines, (size_t)w, w, h, 2, 1, line_len, -1, -1, iterations);
    fr_morphology_ex(thresh, (size_t)w, hlines, (size_t)w, w, h, 2, line_len, 1, -1, -1, iterations);
    fr_canny(gray, (size_t)w, edges, (size_t)w, w, h, c_lo, c_hi, NULL);
    if (accum) fr_hough_accum(edges, (size_t)w, w, h, 1.0, M_PI / 180, accum, numangle, numrho);
}
