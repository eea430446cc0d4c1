"""A batch of very many tiny images (default 70000 of 24 x 20: more images than a grid dimension holds) through the
fused pass and the batched helpers, a sample of them against the CPU restatement.
usage: python scripts/probe_many_small.py [count]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import imageprocessing_amd as ipa  # noqa: E402
from oracle import frontref as fr  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
h, w = 20, 24
rng = np.random.default_rng(77)
bgr = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
bgr[:, 5:15, 3:20] = (bgr[:, 5:15, 3:20] // 4)  # a dark block so that thresholds and opens have something to keep
res = ipa.front_end_batch(bgr, thresh_val=128, vert_len=3, horiz_len=3, iterations=1, hough=True, hough_threshold=6)
torch.cuda.synchronize()
acc = res.accum.cpu().numpy() if isinstance(res.accum, torch.Tensor) else res.accum
bad = 0
for i in list(range(0, 40)) + list(range(65500, min(n, 65580))) + list(range(max(0, n - 40), n)):
    g = fr.bgr2gray(bgr[i])
    t = fr.threshold(g, 128, 255, 1)[1]
    e = fr.canny(g, 50, 150)
    ok = (np.array_equal(np.asarray(res.gray[i]), g) and np.array_equal(np.asarray(res.thresh[i]), t)
          and np.array_equal(np.asarray(res.vlines[i]), fr.morphology_ex(t, 2, (1, 3), 1))
          and np.array_equal(np.asarray(res.hlines[i]), fr.morphology_ex(t, 2, (3, 1), 1))
          and np.array_equal(np.asarray(res.edges[i]), e) and np.array_equal(acc[i], fr.hough_accum(e, 1.0, np.pi / 180)))
    if not ok:
        bad += 1
        which = [k for k, a, b in (("gray", res.gray[i], g), ("thresh", res.thresh[i], t),
                                   ("vlines", res.vlines[i], fr.morphology_ex(t, 2, (1, 3), 1)),
                                   ("hlines", res.hlines[i], fr.morphology_ex(t, 2, (3, 1), 1)), ("edges", res.edges[i], e),
                                   ("accum", acc[i], fr.hough_accum(e, 1.0, np.pi / 180))) if not np.array_equal(np.asarray(a), b)]
        if bad <= 6 or i > n - 3:
            print("image", i, "differs in", which, flush=True)
segs = ipa.segments_batch(np.asarray(res.hlines[:70000 if n >= 70000 else n]))
for i in (0, 1, len(segs) - 1):
    want = fr.external_boxes(np.asarray(res.hlines[i]))
    if list(map(tuple, want)) != list(map(tuple, segs[i])):
        bad += 1
        print("segments of image", i, "differ", flush=True)
print("many small images: %d images, %d bad" % (n, bad), flush=True)
sys.exit(1 if bad else 0)
