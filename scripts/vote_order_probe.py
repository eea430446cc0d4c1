"""Upper bound of what a different order of the edge-pixel lists could give the vote kernel:
times the vote on the lists as compacted and on the same lists randomly permuted per image."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imageprocessing_amd as ipa
from imageprocessing_amd import _native
from imageprocessing_amd._device import DevImage, stream_ptr, workspace
from imageprocessing_amd.synth import synth_batch

n, h, w = 64, 2160, 3840
L = _native.lib()
host = synth_batch(2000, 8, h, w)
bgr = DevImage.empty(n, h, w, 3)
for i in range(n):
    bgr.t[i, :, : w * 3] = torch.from_numpy(host[i % 8].reshape(h, w * 3)).cuda()
gray, edges = DevImage.empty(n, h, w, 1), DevImage.empty(n, h, w, 1)
st = stream_ptr()
def chk(rc):
    if rc: raise RuntimeError(_native.last_error())
chk(L.ipa_bgr2gray_u8(bgr.ptr, bgr.pitch, bgr.stride, gray.ptr, gray.pitch, gray.stride, w, h, n, st))
wsc = workspace(L.ipa_canny_workspace_bytes(w, h, n))
chk(L.ipa_canny_u8(gray.ptr, gray.pitch, gray.stride, edges.ptr, edges.pitch, edges.stride, w, h, n, 50.0, 150.0, wsc.data_ptr(), wsc.numel(), st))
na, nr = ipa.hough_dims(w, h, 1.0, np.pi / 180)
acc = torch.empty((n, na + 2, nr + 2), dtype=torch.int32, device="cuda")
seg = torch.empty((n, na, L.ipa_hough_segments(nr)), dtype=torch.int32, device="cuda")
cnt = torch.zeros(n, dtype=torch.int32, device="cuda")
mp = w * h // 4
ws = torch.empty(L.ipa_hough_workspace_bytes(w, h, n, mp), dtype=torch.uint8, device="cuda")
def hough():
    chk(L.ipa_hough_accum_u8(edges.ptr, edges.pitch, edges.stride, w, h, n, 1.0, np.pi / 180, acc.data_ptr(), acc.stride(0), seg.data_ptr(), mp, cnt.data_ptr(), ws.data_ptr(), ws.numel(), st))
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
t_all = timed(hough)
ref = acc.clone()
os.environ["IPA_VOTE_ONLY"] = "1"
t_vote = timed(hough)
assert torch.equal(acc, ref)
stride = (mp * 4 + 255) // 256 * 256 // 4
lists = ws[: n * stride * 4].view(torch.int32).view(n, stride)
c = cnt.cpu().numpy()
for i in range(n):
    k = int(c[i])
    lists[i, :k] = lists[i, :k][torch.randperm(k, device="cuda")]
t_perm = timed(hough)
assert torch.equal(acc, ref)
# raster order for comparison
for i in range(n):
    k = int(c[i])
    lists[i, :k] = torch.sort(lists[i, :k])[0]
t_raster = timed(hough)
assert torch.equal(acc, ref)
print(json.dumps({"compact+vote_ms": round(t_all, 4), "vote_as_compacted_ms": round(t_vote, 4), "vote_random_order_ms": round(t_perm, 4), "vote_raster_order_ms": round(t_raster, 4), "points_per_image": int(c.mean())}))
