"""End-to-end rate of the batched upload loop (host arrays in, host results out, PCIe included):
process_images on 128 synthetic 1080p diagrams with 1, 2 and 3 device batches in flight."""
import sys, time, io, numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imageprocessing_amd as ipa
from imageprocessing_amd.synth import synth_batch
imgs = list(synth_batch(3000, 16, 1080, 1920)) * 8          # 128 images of 1080p
for ov in (1, 2, 3):
    ipa.process_images(imgs[:32], max_batch=16, overlap=ov)   # warm-up
    torch.cuda.synchronize(); t = time.time()
    out = ipa.process_images(imgs, thresh_val=150, max_batch=16, overlap=ov)
    torch.cuda.synchronize(); dt = time.time() - t
    print("overlap", ov, "%.1f ms for %d images (%.0f images/s)" % (dt * 1e3, len(imgs), len(imgs) / dt), len(out[5].horizontal_segments))
