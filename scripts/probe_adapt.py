"""adaptiveThreshold (Gaussian 11 / 31, mean 31) and GaussianBlur 9x9 on one 16384x16384 gray plane, a few calls each:
a short workload for kernel-trace and counter passes.  usage: python scripts/probe_adapt.py [size]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageprocessing_amd as cv2  # noqa: E402

s = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
g = torch.Generator().manual_seed(5)
gray = torch.randint(0, 256, (s, s), dtype=torch.uint8, generator=g).cuda()
for _ in range(3):
    cv2.adaptiveThreshold(gray, 255, 1, 0, 11, 2)
    cv2.adaptiveThreshold(gray, 255, 1, 0, 31, 2)
    cv2.adaptiveThreshold(gray, 255, 0, 0, 31, 2)
    cv2.GaussianBlur(gray, (9, 9), 0)
torch.cuda.synchronize()
