"""The Canny entry point that allocates nothing (``ipa_canny_static_u8``: scratch in a device
array of the library) against ``ipa_canny_u8`` with a caller's workspace, through the C ABI.
Two calls in a row share the array; the second one is smaller, so stale contents of the first
call lie behind its part of the scratch."""
import numpy as np
import pytest
import torch

from tests.conftest import uml_like

pytestmark = pytest.mark.gpu


def _planes(gray_np):
    from imageprocessing_amd._device import DevImage, upload

    src = upload(gray_np, 1, batched=gray_np.ndim == 3)
    return src, DevImage.empty(src.batch, src.height, src.width, 1), DevImage.empty(src.batch, src.height, src.width, 1)


def _both(L, gray_np, lo, hi):
    from imageprocessing_amd._device import stream_ptr, workspace

    src, d_ws, d_st = _planes(gray_np)
    d_ws.t.fill_(7)
    d_st.t.fill_(9)
    ws = workspace(L.ipa_canny_workspace_bytes(src.width, src.height, src.batch))
    rc = L.ipa_canny_u8(src.ptr, src.pitch, src.stride, d_ws.ptr, d_ws.pitch, d_ws.stride, src.width, src.height, src.batch,
                        float(lo), float(hi), ws.data_ptr(), ws.numel(), stream_ptr())
    assert rc == 0, L.ipa_last_error()
    rc = L.ipa_canny_static_u8(src.ptr, src.pitch, src.stride, d_st.ptr, d_st.pitch, d_st.stride, src.width, src.height,
                               src.batch, float(lo), float(hi), stream_ptr())
    assert rc == 0, L.ipa_last_error()
    torch.cuda.synchronize()
    return d_ws.valid(), d_st.valid()


def test_static_scratch_canny_matches_workspace_canny_twice():
    from imageprocessing_amd import _native
    from oracle import frontref as fr

    L = _native.lib()
    assert L.ipa_canny_static_max_bytes() == 512 * 1024
    # first call: a batch of three 1080p-like pages (weak chains, many flagged tiles)
    big = np.stack([fr.bgr2gray(uml_like(500 + i, 1080, 1920, noise=3)) for i in range(3)])
    a, b = _both(L, big, 50, 150)
    assert torch.equal(a, b)
    assert np.array_equal(b[1].cpu().numpy(), fr.canny(big[1], 50, 150))
    # second call, smaller input: the array still holds the first call's flags
    small = fr.bgr2gray(uml_like(77, 381, 454, noise=4))
    a, b = _both(L, small, 40, 90)
    assert torch.equal(a, b)
    assert np.array_equal(b[0].cpu().numpy(), fr.canny(small, 40, 90))
    assert int((b > 0).sum().item()) > 1000


def test_static_scratch_canny_refuses_what_does_not_fit():
    from imageprocessing_amd import _native
    from imageprocessing_amd._device import DevImage, stream_ptr

    L = _native.lib()
    # 200 images of 3840x2160 need more scratch than the array holds: refused before anything is launched
    assert L.ipa_canny_workspace_bytes(3840, 2160, 200) > L.ipa_canny_static_max_bytes()
    one = DevImage.empty(1, 16, 256, 1)
    rc = L.ipa_canny_static_u8(one.ptr, one.pitch, 4096 * 2160, one.ptr, one.pitch, 4096 * 2160, 3840, 2160, 200, 50.0, 150.0,
                               stream_ptr())
    assert rc == -3 and b"static scratch" in L.ipa_last_error()
    # empty input: nothing to do
    assert L.ipa_canny_static_u8(one.ptr, one.pitch, one.stride, one.ptr, one.pitch, one.stride, 0, 0, 0, 50.0, 150.0,
                                 stream_ptr()) == 0


def test_static_scratch_canny_on_an_image_of_more_than_4096_tiles():
    # the hysteresis of such images runs region by region and is finished by a device kernel: the call still
    # neither allocates nor waits for the stream
    from imageprocessing_amd import _native
    from oracle import frontref as fr

    L = _native.lib()
    g = fr.bgr2gray(uml_like(901, 2176, 8192, noise=6))
    assert (8192 // 128) * (2176 // 32) > 4096
    a, b = _both(L, g, 50, 150)
    assert torch.equal(a, b)
    assert np.array_equal(b[0].cpu().numpy(), fr.canny(g, 50, 150))
