// The C ABI without Python or PyTorch: one synthetic BGR image through ipa_front_end_u8, buffers from
// hipMalloc, results checked against the CPU restatement (oracle/libfrontref.so) when it is linked in.
// build: see tests/c_abi/Makefile        run: ./front_end_c [width height]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/ipamd.h"

#ifdef WITH_ORACLE
extern "C" void fr_front_end(const uint8_t* bgr, int w, int h, int t, int line_len, int iterations, double c_lo, double c_hi,
                             uint8_t* gray, uint8_t* thresh, uint8_t* vlines, uint8_t* hlines, uint8_t* edges, int* accum, int numangle,
                             int numrho);
#endif

#define HIPCK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

int main(int argc, char** argv)
{
    const int w = argc > 2 ? atoi(argv[1]) : 640, h = argc > 2 ? atoi(argv[2]) : 360;
    // a white sheet with a few dark strokes and a box
    std::vector<uint8_t> bgr((size_t)w * h * 3, 255);
    auto px = [&](int x, int y, uint8_t b, uint8_t g, uint8_t r) {
        if (x >= 0 && x < w && y >= 0 && y < h) { uint8_t* p = &bgr[((size_t)y * w + x) * 3]; p[0] = b; p[1] = g; p[2] = r; }
    };
    for (int x = w / 8; x < 7 * w / 8; x++) for (int t = 0; t < 3; t++) { px(x, h / 4 + t, 20, 20, 20); px(x, 3 * h / 4 + t, 40, 10, 60); }
    for (int y = h / 4; y < 3 * h / 4; y++) for (int t = 0; t < 3; t++) { px(w / 8 + t, y, 0, 0, 0); px(7 * w / 8 - t, y, 30, 30, 30); }
    for (int k = 0; k < h / 3; k++) px(w / 2 + k, h / 3 + k, 10, 60, 10);

    int na = 0, nr = 0;
    if (ipa_hough_dims(w, h, 1.0, M_PI / 180, &na, &nr)) { fprintf(stderr, "%s\n", ipa_last_error()); return 2; }
    const size_t in_pitch = ipa_pitch_for((size_t)w * 3), pitch = ipa_pitch_for((size_t)w), plane = pitch * h;
    const size_t acc_n = (size_t)(na + 2) * (nr + 2), ws_bytes = ipa_front_end_workspace_bytes(w, h, 1, 0);
    uint8_t *d_in, *d_planes;
    int32_t *d_acc, *d_cnt;
    void* d_ws;
    HIPCK(hipMalloc((void**)&d_in, in_pitch * h));
    HIPCK(hipMalloc((void**)&d_planes, plane * 5));
    HIPCK(hipMalloc((void**)&d_acc, acc_n * 4));
    HIPCK(hipMalloc((void**)&d_cnt, 4));
    HIPCK(hipMalloc(&d_ws, ws_bytes));
    HIPCK(hipMemcpy2D(d_in, in_pitch, bgr.data(), (size_t)w * 3, (size_t)w * 3, h, hipMemcpyHostToDevice));
    hipStream_t st;
    HIPCK(hipStreamCreate(&st));
    const int rc = ipa_front_end_u8(d_in, in_pitch, in_pitch * h, d_planes, d_planes + plane, d_planes + 2 * plane, d_planes + 3 * plane,
                                    d_planes + 4 * plane, pitch, plane, d_acc, acc_n, nullptr, 0, d_cnt, w, h, 1, 128, 15, 15, 2, 50.0,
                                    150.0, 0, d_ws, ws_bytes, st);
    if (rc) { fprintf(stderr, "ipa_front_end_u8: %s (%d)\n", ipa_last_error(), rc); return 1; }
    HIPCK(hipStreamSynchronize(st));
    std::vector<uint8_t> out[5];
    for (int k = 0; k < 5; k++) {
        out[k].resize((size_t)w * h);
        HIPCK(hipMemcpy2D(out[k].data(), w, d_planes + k * plane, pitch, w, h, hipMemcpyDeviceToHost));
    }
    std::vector<int32_t> acc(acc_n);
    HIPCK(hipMemcpy(acc.data(), d_acc, acc_n * 4, hipMemcpyDeviceToHost));
    int32_t npts = 0;
    HIPCK(hipMemcpy(&npts, d_cnt, 4, hipMemcpyDeviceToHost));
    long long votes = 0;
    int best = 0;
    for (int32_t v : acc) { votes += v; if (v > best) best = v; }
    size_t nedge = 0;
    for (uint8_t v : out[4]) nedge += v != 0;
    printf("%dx%d: %zu edge pixels (list %d), %lld votes (= 180 x edge pixels: %s), strongest line %d votes\n", w, h, nedge, npts, votes,
           votes == 180LL * (long long)nedge ? "yes" : "NO", best);
    int bad = votes != 180LL * (long long)nedge || (size_t)npts != nedge;
#ifdef WITH_ORACLE
    std::vector<uint8_t> ref[5];
    for (auto& r : ref) r.resize((size_t)w * h);
    std::vector<int32_t> racc(acc_n);
    fr_front_end(bgr.data(), w, h, 128, 15, 2, 50.0, 150.0, ref[0].data(), ref[1].data(), ref[2].data(), ref[3].data(), ref[4].data(),
                 racc.data(), na, nr);
    const char* names[5] = {"gray", "thresh", "vlines", "hlines", "edges"};
    for (int k = 0; k < 5; k++) {
        const bool same = out[k] == ref[k];
        printf("%-6s %s\n", names[k], same ? "identical to the CPU restatement" : "DIFFERS");
        bad |= !same;
    }
    const bool same = acc == racc;
    printf("accum  %s\n", same ? "identical to the CPU restatement" : "DIFFERS");
    bad |= !same;
#endif
    (void)hipFree(d_in); (void)hipFree(d_planes); (void)hipFree(d_acc); (void)hipFree(d_cnt); (void)hipFree(d_ws);
    return bad ? 1 : 0;
}
