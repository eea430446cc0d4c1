"""Batched form of the service's upload loop (receiveFile.py:22-75 handles the uploaded files one
by one: imdecode -> gray -> threshold -> extract_lines -> get_segments per file).

Here the files are decoded on a pool of host threads, images of equal size are grouped, every
group makes one fused device pass (front_end_batch) and one segment pass per line mask, and the
caller gets the per-image results back in upload order for its downstream (contour / OCR) logic.
Decoding stays on the host (PIL): palette and gray images are expanded, alpha is dropped, channel
order is BGR, like cv2.imdecode(..., IMREAD_COLOR) does for 8-bit files; PNG is lossless either
way, JPEG pixels depend on the decoder library.
"""
import io
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np


@dataclass
class ParsedImage:
    """What load_image_from_file + threshold_image + extract_lines + get_segments x2
    (umlClassdiagram.py:12-36) hand to the rest of a parser."""
    image: np.ndarray                      # uint8[H,W,3] BGR
    gray: np.ndarray
    thresh: np.ndarray
    vlines: np.ndarray
    hlines: np.ndarray
    vertical_segments: Optional[List[Tuple[int, int, int, int]]] = None
    horizontal_segments: Optional[List[Tuple[int, int, int, int]]] = None


def decode_bgr(blob) -> Optional[np.ndarray]:
    """bytes / file-like -> uint8[H,W,3] BGR, or None when the data is no image (cv2.imdecode's None).

    Like cv2.imdecode(..., IMREAD_COLOR): palette and gray images are expanded, alpha is dropped, the
    EXIF orientation is applied, and 16-bit samples are reduced to their high byte (cv2 scales 16-bit
    files by 1/256 when it converts them to 8 bits).  Pinned by a cv2-written file: the palette-PNG case
    (tests/golden/output_bmp_pin.npz).  Not pinned by any file of the reference: EXIF orientation and
    16-bit samples."""
    from PIL import Image

    try:
        data = blob.read() if hasattr(blob, "read") else blob
        with Image.open(io.BytesIO(bytes(data))) as im:
            im.load()
            if im.getexif().get(0x0112, 1) not in (0, 1):  # an orientation tag that asks for a transposition
                try:
                    from PIL import ImageOps

                    im = ImageOps.exif_transpose(im)
                except Exception:  # some Pillow versions cannot transpose every mode: keep the stored orientation
                    pass
            if im.mode.startswith("I;16") or im.mode == "I":
                # mode "I" is 32-bit signed (also what older Pillow reports for 16-bit PNG): clip to the 16-bit range
                g = (np.clip(np.asarray(im).astype(np.int64), 0, 65535) >> 8).astype(np.uint8)
                rgb = np.repeat(g[:, :, None], 3, axis=2)
            else:
                rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    except Exception:
        return None
    return np.ascontiguousarray(rgb[:, :, ::-1])


def decode_many(blobs: Iterable, workers: int = 8) -> List[Optional[np.ndarray]]:
    """Decode on `workers` host threads (PIL releases the GIL while it decompresses); order is kept."""
    blobs = list(blobs)
    if not blobs:
        return []
    if workers <= 1 or len(blobs) == 1:
        return [decode_bgr(b) for b in blobs]
    with ThreadPoolExecutor(max_workers=min(int(workers), len(blobs))) as pool:
        return list(pool.map(decode_bgr, blobs))


def group_by_shape(images: Sequence[Optional[np.ndarray]], max_batch: int = 256) -> Dict[Tuple[int, int], List[List[int]]]:
    """Indices of the decodable images grouped by (height, width), every group cut into device
    batches of at most `max_batch` images, upload order kept inside a batch."""
    if max_batch < 1:
        raise ValueError("max_batch must be >= 1")
    by_shape: Dict[Tuple[int, int], List[int]] = {}
    for i, im in enumerate(images):
        if im is None:
            continue
        if im.ndim != 3 or im.shape[2] != 3 or im.dtype != np.uint8:
            raise ValueError(f"image {i}: uint8[H,W,3] expected")
        by_shape.setdefault((int(im.shape[0]), int(im.shape[1])), []).append(i)
    return {k: [v[o : o + max_batch] for o in range(0, len(v), max_batch)] for k, v in by_shape.items()}


def _run_group(images, idx, thresh_val, vert_len, horiz_len, iterations, segments, stream):
    import torch

    from . import _device, pipeline

    n = len(idx)
    h, w = images[idx[0]].shape[:2]
    with torch.cuda.stream(stream):
        # page-locked staging both ways; the planes stay on the device between the fused pass and the
        # segment passes and come back in one copy
        pin = torch.empty((n, h, w, 3), dtype=torch.uint8, pin_memory=True)
        np.stack([images[i] for i in idx], out=pin.numpy())
        dev_in = pin.to(_device._device(), non_blocking=True)
        # the fused pass also leaves the two line masks as 1-bit planes, which is what the segment passes read
        fe = pipeline.FrontEnd(n, h, w, thresh_val=thresh_val, vert_len=vert_len, horiz_len=horiz_len, iterations=iterations,
                               hough=False, edges=False, line_bits=bool(segments))
        fe.run(_device.upload(dev_in, 3, batched=True))
        back = torch.empty((4, n, h, w), dtype=torch.uint8, pin_memory=True)
        for k in range(4):
            back[k].copy_(fe.plane(k).valid(), non_blocking=True)
        vseg = fe.line_segments(0) if segments else [None] * n
        hseg = fe.line_segments(1) if segments else [None] * n
        stream.synchronize()
    host = back.numpy()
    return [(i, ParsedImage(images[i], host[0, k], host[1, k], host[2, k], host[3, k], vseg[k], hseg[k])) for k, i in enumerate(idx)]


def process_images(images: Sequence[Optional[np.ndarray]], thresh_val: int = 128, vert_len: int = 15, horiz_len: int = 15,
                   iterations: int = 2, segments: bool = True, max_batch: int = 256, overlap: int = 3) -> List[Optional[ParsedImage]]:
    """The parsers' prefix for many images at once.  ``thresh_val``: 150 class, 128 sequence /
    activity, 100 state diagrams (umlClassdiagram.py:19, umlSequencediagram.py:22,
    umlActivitydiagram.py:34, umlStateDiagram.py:19).  Undecodable entries stay None.

    ``overlap`` device batches are in flight at once, each on a stream of its own and driven by a
    host thread of its own, so that the upload of one batch, the kernels of another and the
    download of a third share the PCIe link and the GPU (a lone batch is bound by its copies)."""
    import torch

    out: List[Optional[ParsedImage]] = [None] * len(images)
    work = [idx for _, batches in group_by_shape(images, max_batch).items() for idx in batches]
    if not work:
        return out
    args = (thresh_val, vert_len, horiz_len, iterations, segments)
    if overlap <= 1 or len(work) == 1:
        done = [_run_group(images, idx, *args, torch.cuda.current_stream()) for idx in work]
    else:
        from . import _device

        dev = _device._device()
        streams = [torch.cuda.Stream(device=dev) for _ in range(min(int(overlap), len(work)))]
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())

        import queue

        free = queue.Queue()
        for st in streams:
            free.put(st)

        def job(k):
            st = free.get()  # one batch per stream at a time
            try:
                with torch.cuda.device(dev):
                    r = _run_group(images, work[k], *args, st)
                    st.synchronize()
                    return r
            finally:
                free.put(st)

        with ThreadPoolExecutor(max_workers=len(streams)) as pool:
            done = list(pool.map(job, range(len(work))))
    for group in done:
        for i, parsed in group:
            out[i] = parsed
    return out


def process_files(blobs: Iterable, workers: int = 8, **kw) -> List[Optional[ParsedImage]]:
    """decode_many + process_images: the whole upload loop of receiveFile.py:22 up to get_segments."""
    return process_images(decode_many(blobs, workers), **kw)
