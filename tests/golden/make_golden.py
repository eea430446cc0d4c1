"""Builds the committed fixtures under tests/golden/ from DATA files of the reference.

Run once in the build container (``python tests/golden/make_golden.py``); the
reference tree does not travel to the GPU box, the fixtures do.

Origins (every expected value is labelled):
* ``gray_kat.npz``     -- origin ``cv2-artefact``.  NewVersion2/inverted.png and
  NewVersion2/gray_img.png were written by real cv2 from one (un-shipped) input
  (NewVersion2/text_detectionv2.py:22-30): ``inverted = bitwise_not(bgr)``,
  ``gray = cvtColor(bgr, BGR2GRAY)``.  ``255 - inverted`` recovers the BGR input
  (PNG is lossless; PIL reads RGB so the channel order is reversed back).  The
  fixture keeps the 256 rows x 320 cols window holding the most coloured pixels
  plus every distinct colour of the full image with its gray value.
* ``cd.npz`` / ``login.npz`` -- decoded BGR pixels of images/cd.png (the
  config-0 input) and images/login.png (what lineDetection.py:5 loads); inputs
  only, ``restatement`` outputs are derived in the tests.
* ``login.png`` + ``output_bmp_pin.npz`` -- origin ``cv2-artefact``.  output.bmp at the root of the
  reference is ``cv2.imwrite("output.bmp", cv2.imread("images/login.png"))``
  (NewVersion2/img2UML.py:16-17): BMP stores the BGR pixels cv2 decoded, so it pins what
  ``cv2.imread`` / ``cv2.imdecode(IMREAD_COLOR)`` makes of that palette PNG.  The fixture keeps the
  PNG file itself (a 15 kB data file), the SHA-256 of output.bmp's pixel array and a 96x128 crop.
* ``arrow_pin.npz`` -- origin ``cv2-artefact``.  thresh_image.png / arrow_image.png at the root of the
  reference (2000 x 1480, values {0, 255}) were written by newVersion3/detect_arrowv2.py:126,146 (an earlier run
  on an un-shipped input): ``arrow`` = the filled contours (RETR_TREE, drawContours thickness -1, :13,33) of
  those components of ``cv2.dilate(thresh, getStructuringElement(MORPH_RECT, (2, 2)))`` (:10-11) that have a
  deep convexity defect.  Whatever the selection was, every drawn component is a whole 8-connected component of
  the dilated image with its holes filled, so
  ``fill_holes(components of dilate2x2(thresh) that meet arrow) == arrow`` must hold pixel for pixel.  It does
  with the window offsets {-1, 0} (anchor (1, 1) and ignored out-of-image samples); with offsets {0, +1} 9002
  pixels differ.  The pair therefore pins the anchor of an even-sized kernel and the border rule of cv2.dilate.
  The fixture keeps both images whole as packed bits (the invariant is about whole components) with their SHA-256.
* ``images/*.png`` -- the reference's own input diagrams (images/*.png, 41 files, 0.85 MB; the JPEG and the WebP
  are left out), copied as data files: what the scripts and the ``__main__`` blocks of the parsers load by
  hard-coded path.  Inputs only; expected outputs are ``restatement`` values computed in the tests.
"""
import hashlib
import os
import shutil

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_bgr(path):
    return np.asarray(Image.open(path).convert("RGB"))[..., ::-1].copy()


def main():
    inv_rgb = np.asarray(Image.open(f"{REF}/NewVersion2/inverted.png").convert("RGB"))
    gray = np.asarray(Image.open(f"{REF}/NewVersion2/gray_img.png").convert("L"))
    inv_bgr = inv_rgb[..., ::-1]
    bgr = 255 - inv_bgr
    coloured = (bgr.max(axis=2) != bgr.min(axis=2)).astype(np.int64)
    H, W = gray.shape
    ch, cw = 256, 320
    ii = np.pad(coloured.cumsum(0).cumsum(1), ((1, 0), (1, 0)))
    best, by, bx = -1, 0, 0
    for y in range(0, H - ch + 1, 8):
        for x in range(0, W - cw + 1, 8):
            s = ii[y + ch, x + cw] - ii[y, x + cw] - ii[y + ch, x] + ii[y, x]
            if s > best:
                best, by, bx = s, y, x
    flat = np.concatenate([bgr.reshape(-1, 3), gray.reshape(-1, 1)], axis=1)
    uniq = np.unique(flat, axis=0)
    np.savez_compressed(
        f"{OUT}/gray_kat.npz",
        bgr=bgr[by : by + ch, bx : bx + cw].copy(),
        inverted_bgr=inv_bgr[by : by + ch, bx : bx + cw].copy(),
        gray=gray[by : by + ch, bx : bx + cw].copy(),
        colours_bgr=uniq[:, :3].astype(np.uint8),
        colours_gray=uniq[:, 3].astype(np.uint8),
    )
    print("gray KAT window", (by, bx), "coloured px", best, "distinct colour->gray rows", len(uniq))
    for name in ("cd", "login"):
        a = load_bgr(f"{REF}/images/{name}.png")
        np.savez_compressed(f"{OUT}/{name}.npz", bgr=a)
        print(name, a.shape)
    shutil.copyfile(f"{REF}/images/login.png", f"{OUT}/login.png")
    bmp = load_bgr(f"{REF}/output.bmp")
    np.savez_compressed(f"{OUT}/output_bmp_pin.npz", shape=np.array(bmp.shape), sha256=np.frombuffer(
        hashlib.sha256(np.ascontiguousarray(bmp).tobytes()).digest(), np.uint8), crop=bmp[300:396, 500:628].copy())
    print("output.bmp", bmp.shape, hashlib.sha256(np.ascontiguousarray(bmp).tobytes()).hexdigest())
    th = np.asarray(Image.open(f"{REF}/thresh_image.png").convert("L"))
    ar = np.asarray(Image.open(f"{REF}/arrow_image.png").convert("L"))
    assert th.shape == ar.shape and set(np.unique(th)) <= {0, 255} and set(np.unique(ar)) <= {0, 255}
    np.savez_compressed(
        f"{OUT}/arrow_pin.npz", shape=np.array(th.shape),
        thresh_bits=np.packbits(th > 0, axis=1), arrow_bits=np.packbits(ar > 0, axis=1),
        thresh_sha256=np.frombuffer(hashlib.sha256(th.tobytes()).digest(), np.uint8),
        arrow_sha256=np.frombuffer(hashlib.sha256(ar.tobytes()).digest(), np.uint8))
    print("arrow pin", th.shape, int((th > 0).sum()), int((ar > 0).sum()))
    os.makedirs(f"{OUT}/images", exist_ok=True)
    n = 0
    for name in sorted(os.listdir(f"{REF}/images")):
        if name.lower().endswith(".png"):
            shutil.copyfile(f"{REF}/images/{name}", f"{OUT}/images/{name.replace(' - Copy', '_copy')}")
            n += 1
    print("input diagrams", n)


if __name__ == "__main__":
    main()
