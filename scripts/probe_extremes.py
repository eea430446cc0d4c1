"""Edge cases at the limits against the CPU restatement: HoughLinesP over 66000 tiny edge maps in one call, images of
1 x 65535, 65535 x 1, 2 x 65535, 65535 x 3 and 7 x 40000 pixels through the fused pass and the single operations with
the longest windows.  usage: python scripts/probe_extremes.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import imageprocessing_amd as cv2
from oracle import frontref as fr
bad = 0
# 1. HoughLinesP over 66000 tiny edge maps
n, h, w = 66000, 16, 24
rng = np.random.default_rng(5)
e = np.zeros((n, h, w), np.uint8)
e[:, 8, 2:22] = 255
e[np.arange(n), rng.integers(0, h, n), rng.integers(0, w, n)] = 255
res = cv2.hough_lines_p_batch(e, 1, np.pi / 180, 8, minLineLength=5, maxLineGap=1)
for i in (0, 1, 65534, 65535, 65536, n - 1):
    want = fr.hough_lines_p(e[i], 1, np.pi / 180, 8, 5, 1)
    got = res[i]
    ok = (want is None and got is None) or (want is not None and got is not None and np.array_equal(got, want))
    if not ok:
        bad += 1
        print("HoughLinesP image", i, "differs", got, want)
print("HoughLinesP 66000 images done", flush=True)
# 4. extreme aspect ratios through the fused pass
for (hh, ww) in ((1, 65535), (65535, 1), (2, 65535), (65535, 3), (7, 40000)):
    img = rng.integers(0, 256, (hh, ww, 3), dtype=np.uint8)
    r = cv2.front_end(img, thresh_val=128, vert_len=15, horiz_len=15, iterations=2, hough_threshold=50)
    g = fr.bgr2gray(img)
    t = fr.threshold(g, 128, 255, 1)[1]
    ed = fr.canny(g, 50, 150)
    oks = [np.array_equal(r.gray, g), np.array_equal(r.thresh, t), np.array_equal(r.vlines, fr.morphology_ex(t, 2, (1, 15), 2)),
           np.array_equal(r.hlines, fr.morphology_ex(t, 2, (15, 1), 2)), np.array_equal(r.edges, ed),
           np.array_equal(r.accum.cpu().numpy(), fr.hough_accum(ed, 1.0, np.pi / 180))]
    print((hh, ww), oks, flush=True)
    bad += 0 if all(oks) else 1
    # single ops on the gray plane
    gt = torch.from_numpy(g).cuda()
    for name, got, want in (("blur9", cv2.GaussianBlur(gt, (9, 9), 0), fr.gaussian_blur(g, (9, 9), 0)),
                            ("median5", cv2.medianBlur(gt, 5), fr.median_blur(g, 5)),
                            ("adaptive31", cv2.adaptiveThreshold(gt, 255, 1, 0, 31, 2), fr.adaptive_threshold(g, 255, 1, 0, 31, 2)),
                            ("erode 1x101", cv2.erode(gt, np.ones((101, 1), np.uint8)), fr.morphology_ex(g, 0, (1, 101), 1)),
                            ("dilate 101x1", cv2.dilate(gt, np.ones((1, 101), np.uint8)), fr.morphology_ex(g, 1, (101, 1), 1)),
                            ("segments", cv2.get_segments(torch.from_numpy(t).cuda()), [tuple(b) for b in fr.external_boxes(t)])):
        if name == "segments":
            ok = [tuple(b) for b in got] == want
        else:
            ok = np.array_equal(got.cpu().numpy(), want)
        if not ok:
            bad += 1
            print("  ", (hh, ww), name, "differs", flush=True)
print("bad", bad)
sys.exit(1 if bad else 0)
