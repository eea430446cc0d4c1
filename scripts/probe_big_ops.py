"""Single operations on ONE image of more than 2^31 pixels (default 46400 x 46400) against the CPU restatement.
usage: python scripts/probe_big_ops.py [width height]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import imageprocessing_amd as cv2  # noqa: E402
from imageprocessing_amd.synth import synth_diagram  # noqa: E402
from oracle import frontref as fr  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (46400, 46400)
t0 = time.perf_counter()
# a quarter-size diagram tiled 2 x 2 (the generator is the slow part), with paper noise so that gray levels vary
q = synth_diagram(9200, (h + 1) // 2, (w + 1) // 2, noise=6)
gray = np.ascontiguousarray(np.tile(fr.bgr2gray(q), (2, 2))[:h, :w])
del q
print("input %.1f s, %d x %d" % (time.perf_counter() - t0, w, h), flush=True)
g = torch.from_numpy(gray).cuda()
bad = []


def check(name, got, want):
    got = got.cpu().numpy() if isinstance(got, torch.Tensor) else got
    ok = np.array_equal(got, want)
    print("%-34s %s" % (name, "ok" if ok else "MISMATCH (%d bytes)" % int((got != want).sum())), flush=True)
    if not ok:
        bad.append(name)


th = fr.threshold(gray, 128, 255, 1)[1]
check("threshold BINARY_INV", cv2.threshold(g, 128, 255, cv2.THRESH_BINARY_INV)[1], th)
r, o = fr.threshold(gray, 0, 255, 8)
rr, oo = cv2.threshold(g, 0, 255, cv2.THRESH_BINARY | cv2.THRESH_OTSU)
check("threshold OTSU (value %g / %g)" % (rr, r), oo, o)
check("GaussianBlur 5x5", cv2.GaussianBlur(g, (5, 5), 0), fr.gaussian_blur(gray, (5, 5), 0))
check("GaussianBlur 9x9", cv2.GaussianBlur(g, (9, 9), 0), fr.gaussian_blur(gray, (9, 9), 0))
check("medianBlur 3", cv2.medianBlur(g, 3), fr.median_blur(gray, 3))
check("medianBlur 5", cv2.medianBlur(g, 5), fr.median_blur(gray, 5))
for blk, method in ((11, 1), (31, 1), (11, 0)):
    check("adaptiveThreshold %s %d" % ("GAUSSIAN" if method else "MEAN", blk),
          cv2.adaptiveThreshold(g, 255, method, cv2.THRESH_BINARY, blk, 2), fr.adaptive_threshold(gray, 255, method, 0, blk, 2))
t = torch.from_numpy(th).cuda()
for name, src, srcn in (("gray", g, gray), ("binary", t, th)):
    for op, ks, it in ((0, (3, 3), 1), (1, (1, 15), 2), (2, (15, 1), 2), (3, (3, 3), 2), (0, (101, 1), 1), (1, (1, 101), 1)):
        k = np.ones((ks[1], ks[0]), np.uint8)
        got = cv2.morphologyEx(src, op, k, iterations=it) if op >= 2 else (cv2.erode if op == 0 else cv2.dilate)(src, k, iterations=it)
        check("morph %s op %d %dx%d it=%d" % (name, op, ks[0], ks[1], it), got, fr.morphology_ex(srcn, op, ks, it))
check("bitwise_not", cv2.bitwise_not(g), 255 - gray)
print("FAILED: %s" % bad if bad else "all ok", flush=True)
sys.exit(1 if bad else 0)
