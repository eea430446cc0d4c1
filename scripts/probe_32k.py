"""One 32768 x 24576 raster (0.8 gigapixels) through the fused pass with the Hough stage, against the CPU restatement:
the largest single image tried (row offsets close to 2^32, rho range of 81921 bins per angle).
usage: python scripts/probe_32k.py [width height]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import imageprocessing_amd as ipa  # noqa: E402
from imageprocessing_amd.synth import synth_diagram  # noqa: E402
from oracle import frontref as fr  # noqa: E402

w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32768, 24576)
t0 = time.perf_counter()
bgr = synth_diagram(9100, h, w, noise=4)
print("synth %.1f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
res = ipa.front_end(bgr, thresh_val=128, vert_len=15, horiz_len=15, iterations=2, hough_threshold=4000)
torch.cuda.synchronize()
print("device %.2f s host to host" % (time.perf_counter() - t0), flush=True)
gray = fr.bgr2gray(bgr)
th = fr.threshold(gray, 128, 255, 1)[1]
ok = {"gray": np.array_equal(res.gray, gray), "thresh": np.array_equal(res.thresh, th),
      "vlines": np.array_equal(res.vlines, fr.morphology_ex(th, 2, (1, 15), 2)),
      "hlines": np.array_equal(res.hlines, fr.morphology_ex(th, 2, (15, 1), 2))}
e = fr.canny(gray, 50, 150)
ok["edges"] = np.array_equal(res.edges, e)
if not ok["edges"]:
    rows = np.flatnonzero((res.edges != e).any(axis=1))
    d = res.edges[rows[0]] != e[rows[0]]
    print("edges differ in %d rows, first %d last %d; in row %d at columns %s; device %s oracle %s" % (
        len(rows), rows[0], rows[-1], rows[0], np.flatnonzero(d)[:8], res.edges[rows[0]][d][:8], e[rows[0]][d][:8]), flush=True)
    print("differing pixels: %d, device-only edges %d, oracle-only edges %d" % (
        int((res.edges != e).sum()), int(((res.edges != 0) & (e == 0)).sum()), int(((res.edges == 0) & (e != 0)).sum())), flush=True)
acc = fr.hough_accum(e, 1.0, np.pi / 180)
ok["accum"] = np.array_equal(res.accum.cpu().numpy(), acc)
want = fr.hough_peaks(acc, 1.0, np.pi / 180, 4000)
ok["lines"] = (want is None and res.lines is None) or (want is not None and res.lines is not None and np.array_equal(res.lines, want))
print(ok, "lines", None if want is None else len(want), flush=True)
sys.exit(0 if all(ok.values()) else 1)
