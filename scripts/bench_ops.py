"""Per-operation device timings through the C ABI (inputs resident in HBM), one JSON line per op.

usage: python scripts/bench_ops.py [--batch 64] [--size 4k]
Algorithmic bytes per pixel follow the per-op figures of the measurement plan: gray 4, threshold 2,
Otsu 1 + 2, each erode/dilate/open/close 2, GaussianBlur 2, adaptiveThreshold 2, Canny 2, Hough 1 + accumulator.
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", default="4k")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="", help="run only the ops whose name contains this text")
    args = ap.parse_args()
    h, w = {"1080p": (1080, 1920), "4k": (2160, 3840)}[args.size]
    import torch

    import imageprocessing_amd as ipa
    from imageprocessing_amd import _native
    from imageprocessing_amd._device import DevImage, stream_ptr, workspace
    from imageprocessing_amd.synth import synth_batch

    L = _native.lib()
    n = args.batch
    host = synth_batch(2000, min(n, 8), h, w)
    bgr = DevImage.empty(n, h, w, 3)
    for i in range(n):
        bgr.t[i, :, : w * 3] = torch.from_numpy(host[i % host.shape[0]].reshape(h, w * 3)).cuda()
    gray, thr, out = DevImage.empty(n, h, w, 1), DevImage.empty(n, h, w, 1), DevImage.empty(n, h, w, 1)
    st = stream_ptr()

    def chk(rc):
        if rc:
            raise RuntimeError(_native.last_error())

    chk(L.ipa_bgr2gray_u8(bgr.ptr, bgr.pitch, bgr.stride, gray.ptr, gray.pitch, gray.stride, w, h, n, st))
    chk(L.ipa_threshold_u8(gray.ptr, gray.pitch, gray.stride, thr.ptr, thr.pitch, thr.stride, w, h, n, 128.0, 255.0, 1, st))
    hist = torch.zeros((n, 256), dtype=torch.int32, device="cuda")
    otsu = torch.zeros(n, dtype=torch.int32, device="cuda")
    na, nr = ipa.hough_dims(w, h, 1.0, np.pi / 180)
    acc = torch.empty((n, na + 2, nr + 2), dtype=torch.int32, device="cuda")
    seg = torch.empty((n, na, L.ipa_hough_segments(nr)), dtype=torch.int32, device="cuda")
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
    ws_sizes = [L.ipa_adaptive_threshold_workspace_bytes(w, h, n), L.ipa_gaussian_blur_workspace_bytes(w, h, n),
                L.ipa_morph_workspace_bytes(w, h, n), L.ipa_canny_workspace_bytes(w, h, n),
                L.ipa_hough_workspace_bytes(w, h, n, w * h // 4), L.ipa_segments_workspace_bytes(w, h, n, 0)]
    ws = workspace(max(ws_sizes))
    px = float(n) * h * w
    acc_bytes = 4.0 * (na + 2) * (nr + 2) * n
    edges = DevImage.empty(n, h, w, 1)
    chk(L.ipa_canny_u8(gray.ptr, gray.pitch, gray.stride, edges.ptr, edges.pitch, edges.stride, w, h, n, 50.0, 150.0,
                       ws.data_ptr(), ws.numel(), st))

    def morph(op, kw, kh, it, binary, src):
        return lambda: chk(L.ipa_morph_u8(src.ptr, src.pitch, src.stride, out.ptr, out.pitch, out.stride, w, h, n, op, kw, kh,
                                          -1, -1, it, binary, ws.data_ptr(), ws.numel(), st))

    hl = DevImage.empty(n, h, w, 1)
    chk(L.ipa_morph_u8(thr.ptr, thr.pitch, thr.stride, hl.ptr, hl.pitch, hl.stride, w, h, n, 2, 15, 1, -1, -1, 2, 1,
                       ws.data_ptr(), ws.numel(), st))
    boxes = torch.empty((n, 4096, 4), dtype=torch.int32, device="cuda")
    nbox = torch.zeros(n, dtype=torch.int32, device="cuda")
    rects = torch.tensor([[i % n, 100 + 7 * i, 50 + 11 * i, 900 + 7 * i, 54 + 11 * i] for i in range(4 * n)], dtype=torch.int32).cuda()

    def segments(src):
        return lambda: chk(L.ipa_segments_u8(src.ptr, src.pitch, src.stride, w, h, n, 0, boxes.data_ptr(), 4096, nbox.data_ptr(),
                                             ws.data_ptr(), ws.numel(), st))

    hp_ws = workspace(L.ipa_hough_lines_p_workspace_bytes(w, h, n, 1.0, np.pi / 180, w * h // 4))
    hp_lines = torch.empty((n, 4096, 4), dtype=torch.int32, device="cuda")

    ops = [
        ("cvtColor BGR2GRAY", 4 * px, lambda: chk(L.ipa_bgr2gray_u8(bgr.ptr, bgr.pitch, bgr.stride, gray.ptr, gray.pitch, gray.stride, w, h, n, st))),
        ("threshold BINARY_INV", 2 * px, lambda: chk(L.ipa_threshold_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 128.0, 255.0, 1, st))),
        ("threshold OTSU", 3 * px, lambda: chk(L.ipa_threshold_otsu_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 255.0, 0, hist.data_ptr(), otsu.data_ptr(), st))),
        ("adaptiveThreshold GAUSSIAN 11", 2 * px, lambda: chk(L.ipa_adaptive_threshold_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 255.0, 1, 0, 11, 2.0, ws.data_ptr(), ws.numel(), st))),
        ("adaptiveThreshold GAUSSIAN 31", 2 * px, lambda: chk(L.ipa_adaptive_threshold_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 255.0, 1, 0, 31, 2.0, ws.data_ptr(), ws.numel(), st))),
        ("adaptiveThreshold MEAN 11", 2 * px, lambda: chk(L.ipa_adaptive_threshold_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 255.0, 0, 0, 11, 2.0, ws.data_ptr(), ws.numel(), st))),
        ("GaussianBlur 5x5 sigma 0", 2 * px, lambda: chk(L.ipa_gaussian_blur_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 5, 5, 0.0, 0.0, 4, ws.data_ptr(), ws.numel(), st))),
        ("GaussianBlur 5x5 sigma 1", 2 * px, lambda: chk(L.ipa_gaussian_blur_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 5, 5, 1.0, 0.0, 4, ws.data_ptr(), ws.numel(), st))),
        ("GaussianBlur 9x9 sigma 0", 2 * px, lambda: chk(L.ipa_gaussian_blur_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 9, 9, 0.0, 0.0, 4, ws.data_ptr(), ws.numel(), st))),
        ("medianBlur 3", 2 * px, lambda: chk(L.ipa_median_blur_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 3, st))),
        ("medianBlur 5", 2 * px, lambda: chk(L.ipa_median_blur_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 5, st))),
        ("open 1x15 it=2 (binary)", 2 * px, morph(2, 1, 15, 2, 1, thr)),
        ("open 15x1 it=2 (binary)", 2 * px, morph(2, 15, 1, 2, 1, thr)),
        ("close 3x3 it=1 (binary)", 2 * px, morph(3, 3, 3, 1, 1, thr)),
        ("erode 3x3 it=2 (binary)", 2 * px, morph(0, 3, 3, 2, 1, thr)),
        ("dilate 3x3 it=1 (gray)", 2 * px, morph(1, 3, 3, 1, 0, gray)),
        ("open 15x1 it=2 (gray)", 2 * px, morph(2, 15, 1, 2, 0, gray)),
        ("open 1x15 it=2 (gray)", 2 * px, morph(2, 1, 15, 2, 0, gray)),
        ("erode 101x1 (gray)", 2 * px, morph(0, 101, 1, 1, 0, gray)),
        ("erode 1x101 (gray)", 2 * px, morph(0, 1, 101, 1, 0, gray)),
        ("open 101x101 (gray)", 2 * px, morph(2, 101, 101, 1, 0, gray)),
        ("close 3x3 it=2 (gray)", 2 * px, morph(3, 3, 3, 2, 0, gray)),
        ("Canny 50/150", 2 * px, lambda: chk(L.ipa_canny_u8(gray.ptr, gray.pitch, gray.stride, out.ptr, out.pitch, out.stride, w, h, n, 50.0, 150.0, ws.data_ptr(), ws.numel(), st))),
        ("HoughLines accumulator", px + acc_bytes, lambda: chk(L.ipa_hough_accum_u8(edges.ptr, edges.pitch, edges.stride, w, h, n, 1.0, np.pi / 180, acc.data_ptr(), acc.stride(0), seg.data_ptr(), w * h // 4, cnt.data_ptr(), ws.data_ptr(), ws.numel(), st))),
        ("HoughLinesP 100/100/10", px, lambda: chk(L.ipa_hough_lines_p_u8(edges.ptr, edges.pitch, edges.stride, w, h, n, 1.0, np.pi / 180, 100, 100, 10, w * h // 4, hp_lines.data_ptr(), 4096, nbox.data_ptr(), hp_ws.data_ptr(), hp_ws.numel(), st))),
        ("get_segments (hlines mask)", px, segments(hl)),
        ("get_segments (thresh mask)", px, segments(thr)),
        ("erase rects (4 per image)", 0.0, lambda: chk(L.ipa_fill_rects_c3_u8(bgr.ptr, bgr.pitch, bgr.stride, w, h, n, rects.data_ptr(), rects.shape[0], 255, 255, 255, st))),
        ("cvtColor + threshold fused", 5 * px, lambda: chk(L.ipa_gray_threshold_u8(bgr.ptr, bgr.pitch, bgr.stride, gray.ptr, out.ptr, gray.pitch, gray.stride, w, h, n, 128.0, 255.0, 1, st))),
        ("bitwise_or", 3 * px, lambda: chk(L.ipa_bitwise_or_u8(gray.ptr, gray.pitch, gray.stride, thr.ptr, thr.pitch, thr.stride, out.ptr, out.pitch, out.stride, w, h, n, st))),
    ]
    for name, nbytes, fn in ops:
        if args.only and args.only not in name:
            continue
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        print(json.dumps({"op": name, "images": n, "size": f"{w}x{h}", "ms": round(ms, 4), "Mpix_per_s": round(px / ms / 1e3, 1),
                          "algorithmic_GBps": round(nbytes / ms / 1e6, 1), "frac_of_8TBps": round(nbytes / ms / 1e6 / 8000, 4)}), flush=True)


if __name__ == "__main__":
    main()
