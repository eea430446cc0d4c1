"""imageprocessing_amd -- MI355X (gfx950) raster front-end with cv2-style call signatures.

``import imageprocessing_amd as cv2`` covers the calls the reference's diagram
parsers make before their contour logic: cvtColor, threshold (binary / Otsu),
adaptiveThreshold, GaussianBlur, medianBlur, getStructuringElement, erode, dilate,
morphologyEx, Canny, HoughLines, bitwise_or, bitwise_not.  The pixel work runs in
hand-written HIP kernels (``libipamd_hip.so``, C ABI in ``include/ipamd.h``);
PyTorch-ROCm only owns the device buffers.  There is no CPU implementation.
"""
from ._device import DevImage, error  # noqa: F401
from .cv2api import *  # noqa: F401,F403
from .cv2api import hough_accumulator, hough_dims, hough_peaks, hough_peaks_batch, lines_from_keys  # noqa: F401
from .pipeline import (FrontEnd, FrontEndResult, edges_and_lines, erase_rects, front_end, front_end_batch,  # noqa: F401
                       get_option, get_segments, rebinarize, segments_batch, set_option)
from .batching import ParsedImage, decode_bgr, decode_many, group_by_shape, process_files, process_images  # noqa: F401
from .sharding import front_end_sharded, gather_to_rank0, plan_shards, shard_range, shard_round_robin  # noqa: F401

__version__ = "0.1.0"
