"""Kernel-level look at Canny on the config-4 raster under different thresholds (run under rocprofv3 --kernel-trace).
usage: python scripts/probe_hyst16k.py lo hi [noise]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imageprocessing_amd as cv2
from imageprocessing_amd.synth import synth_diagram
lo, hi = float(sys.argv[1]), float(sys.argv[2])
noise = int(sys.argv[3]) if len(sys.argv) > 3 else 6
s = 16384
bgr = torch.from_numpy(synth_diagram(9000, s, s, noise=noise, gradient=True)).cuda()
gray = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
for _ in range(3):
    e = cv2.Canny(gray, lo, hi)
torch.cuda.synchronize()
print("edge pixels", int((e > 0).sum().item()))
