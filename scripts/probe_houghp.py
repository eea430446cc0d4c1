"""HoughLinesP on one 4K edge map and on the edge map of tests/golden/login.png: device time per call.
usage: python scripts/probe_houghp.py   (IPA_HP_STATS=1 prints the phase timers of image 0, IPA_HP_WAVES=1|4|16 the variant)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import imageprocessing_amd as cv2  # noqa: E402
from imageprocessing_amd.synth import synth_diagram  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [("synthetic 4K", synth_diagram(2000, 2160, 3840)),
         ("login.png", np.array(Image.open(os.path.join(ROOT, "tests/golden/login.png")).convert("RGB"))[..., ::-1].copy())]
for name, img in cases:
    g = cv2.cvtColor(torch.from_numpy(img).cuda(), cv2.COLOR_BGR2GRAY)
    e = cv2.Canny(g, 50, 150)
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lines = cv2.HoughLinesP(e, 1, np.pi / 180, 100, minLineLength=100, maxLineGap=10)
        torch.cuda.synchronize()
        print("HoughLinesP %s %s: %.2f ms, %d lines, %d edge px" % (name, tuple(e.shape), (time.perf_counter() - t0) * 1e3,
                                                                  0 if lines is None else len(lines), int((e != 0).sum())))
