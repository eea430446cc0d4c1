"""One Canny call on the config-4 raster with a build of the library that prints the rounds of some regions
(IPA_LIB_PATH selects it)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imageprocessing_amd as cv2
from imageprocessing_amd.synth import synth_diagram
s = 16384
bgr = torch.from_numpy(synth_diagram(9000, s, s, noise=6, gradient=True)).cuda()
gray = cv2.cvtColor(bgr, cv2.COLOR_BGR2GRAY)
e = cv2.Canny(gray, 50, 150)
torch.cuda.synchronize()
