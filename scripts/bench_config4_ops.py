"""Device timings of the blur / Otsu / adaptive / close / open prefix of NewVersion2/img2UML.py:26-35 and of variant B
on the config-4 raster (one 16384x16384 whiteboard scan, seed 9000, noise +-6, gradient): one JSON line per operation.
usage: python scripts/bench_config4_ops.py [--size 16384]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", default="", help="substring of the operations to run")
    args = ap.parse_args()
    import torch

    import imageprocessing_amd as cv2
    from imageprocessing_amd.synth import synth_diagram

    s = args.size
    bgr = torch.from_numpy(synth_diagram(9000, s, s, noise=6, gradient=True)).cuda()
    gray = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
    blur = cv2.GaussianBlur(gray, (5, 5), 0)
    ad = cv2.adaptiveThreshold(blur, 255, cv2.ADAPTIVE_THRESH_GAUSSIAN_C, cv2.THRESH_BINARY, 11, 2)
    k3 = np.ones((3, 3), np.uint8)
    px = float(s) * s
    fe = cv2.FrontEnd(1, s, s, vert_len=101, horiz_len=101, iterations=2, blur=True, hough=True)
    from imageprocessing_amd._device import upload

    dev_bgr = upload(bgr, 3)
    ops = [
        ("cvtColor BGR2GRAY", 4, lambda: cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)),
        ("threshold OTSU (noisy page)", 3, lambda: cv2.threshold(gray, 0, 255, cv2.THRESH_BINARY + cv2.THRESH_OTSU)),
        ("GaussianBlur 5x5 sigma 0", 2, lambda: cv2.GaussianBlur(gray, (5, 5), 0)),
        ("adaptiveThreshold GAUSSIAN 11", 2, lambda: cv2.adaptiveThreshold(blur, 255, 1, 0, 11, 2)),
        ("adaptiveThreshold GAUSSIAN 31", 2, lambda: cv2.adaptiveThreshold(gray, 255, 1, 0, 31, 2)),
        ("adaptiveThreshold MEAN 31", 2, lambda: cv2.adaptiveThreshold(gray, 255, 0, 0, 31, 2)),
        ("close 3x3 it=2 (binary)", 2, lambda: cv2.morphologyEx(ad, cv2.MORPH_CLOSE, k3, iterations=2)),
        ("open 3x3 it=1 (binary)", 2, lambda: cv2.morphologyEx(ad, cv2.MORPH_OPEN, k3, iterations=1)),
        ("open 101x1 (gray)", 2, lambda: cv2.morphologyEx(gray, cv2.MORPH_OPEN, np.ones((1, 101), np.uint8))),
        ("open 1x101 (gray)", 2, lambda: cv2.morphologyEx(gray, cv2.MORPH_OPEN, np.ones((101, 1), np.uint8))),
        ("Canny 50/150", 2, lambda: cv2.Canny(gray, 50, 150)),
        ("full front-end, variant B (blur 5x5), lines 101 it=2, Hough", 8 + 4.0 * 182 * (4 * s + 3) / px, lambda: fe.enqueue(dev_bgr)),
    ]
    for name, bpp, fn in ops:
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
        print(json.dumps({"op": name, "size": f"{s}x{s}", "ms": round(ms, 4), "Mpix_per_s": round(px / ms / 1e3, 1),
                          "algorithmic_GBps": round(bpp * px / ms / 1e6, 1),
                          "note": "device tensors in and out, includes the output allocation of the cv2-style call"}), flush=True)


if __name__ == "__main__":
    main()
