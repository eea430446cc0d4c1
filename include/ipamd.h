/*
 * ipamd.h -- C ABI of libipamd_hip.so, the MI355X (gfx950) raster front-end.
 *
 * The reference (jaculangithub/ImageProcessing) has no FFI layer of its own:
 * its diagram parsers call `cv2.*` free functions on numpy arrays.  Each entry
 * point below is what a binding for one of those cv2 calls would bind; the
 * comment on each names the cv2 call and the reference call sites it replaces.
 * The Python package `imageprocessing_amd` binds them with ctypes and presents
 * the cv2 names on top.
 *
 * Conventions (all entry points):
 *   - Every image pointer is a DEVICE pointer (HIP) to pixel (0,0) of image 0
 *     of a batch; `*_pitch` is the byte distance between rows, `*_stride` the
 *     byte distance between consecutive images of the batch.
 *   - Pointers must be 16-byte aligned; pitches and strides multiples of 16
 *     (ipa_pitch_for() returns a suitable pitch).  Bytes of a destination row
 *     between `width` and the pitch may be overwritten.
 *   - Planes are 8-bit unsigned, one channel, unless "bgr"/"c3" (3 interleaved
 *     channels, B,G,R order).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *     enqueue work and return without synchronising unless stated otherwise.
 *   - `ws`/`ws_bytes` is caller-owned device scratch of at least the size the
 *     matching ipa_*_workspace_bytes() query returns; the library allocates no
 *     device memory.  Scratch contents are undefined after the call.
 *   - Return value 0 = success, negative = IPA_ERR_*; ipa_last_error() gives a
 *     thread-local message.
 *   - Inputs are never modified; src and dst must not overlap unless stated.
 */
#ifndef IPAMD_H
#define IPAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IPA_OK 0
#define IPA_ERR_ARG (-1)      /* bad size / parameter (cv2 would raise cv2.error) */
#define IPA_ERR_ALIGN (-2)    /* pointer or pitch not 16-byte aligned */
#define IPA_ERR_WORKSPACE (-3)/* workspace too small */
#define IPA_ERR_HIP (-4)      /* a HIP runtime call failed */
#define IPA_ERR_OVERFLOW (-5) /* a bounded output list was too small */

/* cv2 constants accepted by the entry points */
#define IPA_THRESH_BINARY 0
#define IPA_THRESH_BINARY_INV 1
#define IPA_MORPH_ERODE 0
#define IPA_MORPH_DILATE 1
#define IPA_MORPH_OPEN 2
#define IPA_MORPH_CLOSE 3
#define IPA_BORDER_CONSTANT 0
#define IPA_BORDER_REPLICATE 1
#define IPA_BORDER_REFLECT 2
#define IPA_BORDER_REFLECT_101 4
#define IPA_ADAPTIVE_THRESH_MEAN_C 0
#define IPA_ADAPTIVE_THRESH_GAUSSIAN_C 1

int ipa_version(void);
const char* ipa_last_error(void);
/* Smallest multiple of 256 that is >= row_bytes: the pitch the library prefers. */
size_t ipa_pitch_for(size_t row_bytes);

/* Run-time switches of the fused front-end's schedule (no reference counterpart: cv2 exposes only
 * cv2.setNumThreads, which the reference never calls).  Options, with the environment variable that
 * gives the default in brackets:
 *   "pipeline_chunks"  [IPA_PIPELINE]    -1 = by batch size (default); 0 or 1 = one chain on the
 *                                        caller's stream; n <= 8 = that many chunks whose stages
 *                                        overlap on helper streams
 *   "overlap"          [IPA_OVERLAP]     0 = a single chain keeps the opens on the caller's stream
 *   "phased"           [IPA_PHASED]      0 = chunks alternate between two streams also with a Hough stage
 *   "head_start_ticks" [IPA_HEAD_START]  head start of a resident vote kernel in 10 ns ticks
 *   "canny_bands"      [IPA_CANNY_BANDS] rows per band of the Canny strip kernel: -1 = by launch size
 *                                        (default), 16 or 64 = forced
 *   "canny_queue"      [IPA_CANNY_QUEUE] 0 = non-maximum suppression in place in every row instead
 *                                        of deferred through a per-wave LDS stack (64-row bands)
 *   "canny_wide"       [IPA_CANNY_WIDE]  0 = strip kernel (default); 4 or 8 = the Canny kernel that writes whole
 *                                        aligned lines, with that many pixel pairs per lane (fused with the BGR
 *                                        conversion inside ipa_front_end_u8); -1 = that kernel for large launches
 *   "vote_gray"        [IPA_VOTE_GRAY]   0 = default; 1 = in the phased schedule the gray conversion of the next
 *                                        chunk runs on spare waves of the vote kernel; 2 = as 1, opens one phase later
 *   "hyst_rounds"      [IPA_HYST_ROUNDS] Canny on images of more than 4096 tiles: region launches of the hysteresis
 *                                        before the kernel that finishes on the device (default 6)
 * Results do not depend on any of them.  Settings apply to the calls that follow, from any thread. */
int ipa_set_option(const char* name, int value);
int ipa_get_option(const char* name, int* value);

/* Optional per-stage timing of the fused front-end (no reference counterpart:
 * the reference only reports whole-request wall time, umlClassdiagram.py:1096,
 * 1145-1151).  When enabled, HIP events are recorded on the caller's stream around
 * every stage; ipa_profile_collect() waits for them and returns, per stage, the
 * summed milliseconds and the number of timed intervals since the last collect.
 * Arrays must hold ipa_profile_stage_count() entries.  While timing is enabled the
 * fused front-end runs its stages strictly one after the other on the caller's
 * stream (otherwise it overlaps independent stages on helper streams, which makes
 * per-stage durations meaningless). */
int ipa_profile_enable(int on);
int ipa_profile_stage_count(void);
const char* ipa_profile_stage_name(int stage);
int ipa_profile_collect(double* ms, int* launches);

/* ---- colour conversion ------------------------------------------------- */

/* cv2.cvtColor(img, cv2.COLOR_BGR2GRAY): umlClassdiagram.py:14,1121;
 * umlSequencediagram.py:17,605; umlActivitydiagram.py:26,375,425;
 * umlStateDiagram.py:14,360; lineDetection.py:6; shapeDetection.py:7. */
int ipa_bgr2gray_u8(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride,
                    uint8_t* gray, size_t gray_pitch, size_t gray_stride,
                    int width, int height, int batch, void* stream);

/* cv2.cvtColor(gray, cv2.COLOR_GRAY2BGR): umlClassdiagram.py:1103. */
int ipa_gray2bgr_u8(const uint8_t* gray, size_t gray_pitch, size_t gray_stride,
                    uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride,
                    int width, int height, int batch, void* stream);

/* cv2.cvtColor(img, cv2.COLOR_RGB2BGR / COLOR_BGR2RGB):
 * NewVersion2/text_detectionv2.py:17. */
int ipa_swap_rb_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                   uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                   int width, int height, int batch, void* stream);

/* cv2.bitwise_not: NewVersion2/text_detectionv2.py:22.  row_bytes = width*channels. */
int ipa_bitwise_not_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                       uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                       int row_bytes, int height, int batch, void* stream);

/* cv2.bitwise_or: "New Version/textDetection copy.py":94. */
int ipa_bitwise_or_u8(const uint8_t* a, size_t a_pitch, size_t a_stride,
                      const uint8_t* b, size_t b_pitch, size_t b_stride,
                      uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                      int row_bytes, int height, int batch, void* stream);

/* image_rect[y1:y2, x1:x2] = bg_color, the erase step of draw_rectangles
 * (umlClassdiagram.py:107-116, 1117; umlSequencediagram.py:108-141): fills `count`
 * rectangles of one colour in a batch of BGR images IN PLACE, so that the images stay
 * resident for the second binarisation.  `rects` = int32[count][5] in DEVICE memory:
 * image index, x1, y1, x2, y2 with numpy slice semantics (negative bounds count from
 * the end, bounds are clipped, empty ranges do nothing; rectangles of images outside
 * the batch are ignored). */
int ipa_fill_rects_c3_u8(uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride,
                         int width, int height, int batch,
                         const int32_t* rects, int count, int c0, int c1, int c2, void* stream);

/* threshold_image(cv2.cvtColor(img, COLOR_BGR2GRAY)) in one pass over the BGR input: the
 * re-binarisation after the erase (umlClassdiagram.py:1120-1121; umlSequencediagram.py:605)
 * and the first step of every parser (umlClassdiagram.py:14-20).  gray and thresh share
 * one pitch / stride; `type` is IPA_THRESH_BINARY or IPA_THRESH_BINARY_INV. */
int ipa_gray_threshold_u8(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride,
                          uint8_t* gray, uint8_t* thresh, size_t plane_pitch, size_t plane_stride,
                          int width, int height, int batch,
                          double thresh_val, double maxval, int type, void* stream);

/* ---- thresholds -------------------------------------------------------- */

/* cv2.threshold(gray, thresh, maxval, THRESH_BINARY | THRESH_BINARY_INV):
 * umlClassdiagram.py:19; umlSequencediagram.py:22; umlActivitydiagram.py:34;
 * umlStateDiagram.py:19; shapeDetection.py:10; testing.py:50;
 * "New Version/imageToBlackandWhite.py":12.  The value cv2 returns is
 * floor(thresh). */
int ipa_threshold_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                     uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                     int width, int height, int batch,
                     double thresh, double maxval, int type, void* stream);

/* 256-bin histogram per image (uint32[batch][256], zeroed by the call); the
 * first half of cv2.threshold(..., THRESH_OTSU):
 * "New Version/textDetection.py":23. */
int ipa_histogram_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                     int width, int height, int batch, uint32_t* hist, void* stream);

/* cv2.threshold(img, _, maxval, type | THRESH_OTSU): histogram, Otsu scan in
 * double precision and the threshold itself, all on the device.
 * `hist` = uint32[batch][256] scratch/output, `otsu` = int32[batch] output
 * (the value cv2 returns as retval). */
int ipa_threshold_otsu_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                          uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                          int width, int height, int batch,
                          double maxval, int type, uint32_t* hist, int32_t* otsu, void* stream);

/* cv2.adaptiveThreshold(src, maxValue, method, type, blockSize, C):
 * NewVersion2/img2UML.py:29; newVersion3/detect_arrow.py:19;
 * "Version4 with Chatgpt/textDetection.py":15. */
size_t ipa_adaptive_threshold_workspace_bytes(int width, int height, int batch);
int ipa_adaptive_threshold_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                              uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                              int width, int height, int batch,
                              double max_value, int method, int type, int block_size, double c,
                              void* ws, size_t ws_bytes, void* stream);

/* ---- smoothing --------------------------------------------------------- */

/* cv2.GaussianBlur(gray, (kw,kh), sigmaX, sigmaY=0, borderType):
 * NewVersion2/img2UML.py:26; newVersion3/detect_arrow.py:16;
 * newVersion3/detectHorizontalVerticalLines.py:58;
 * "Version4 with Chatgpt/contour_detection.py":13; "images/with API/withAPI.py":50;
 * "New Version/detectArrow.py":7.  8-bit fixed-point arithmetic as OpenCV's. */
size_t ipa_gaussian_blur_workspace_bytes(int width, int height, int batch);
int ipa_gaussian_blur_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                         uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                         int width, int height, int batch,
                         int kw, int kh, double sigma_x, double sigma_y, int border,
                         void* ws, size_t ws_bytes, void* stream);

/* cv2.medianBlur(gray, ksize) for ksize 3 and 5 on 8-bit single-channel images:
 * "New Version/textDetection.py":16 (3, in front of the Otsu threshold of :23),
 * "New Version/detectCircle.py":14, "New Version/detectShape.py":11 (5).
 * Median of the ksize x ksize window, rows and columns outside the image replaced
 * by the nearest one inside.  src and dst must differ.  No workspace. */
int ipa_median_blur_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                       uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                       int width, int height, int batch, int ksize, void* stream);

/* ---- morphology -------------------------------------------------------- */

/* cv2.erode / cv2.dilate / cv2.morphologyEx(src, op, kernel, iterations) with an
 * all-ones kw x kh kernel (cv2.getStructuringElement(MORPH_RECT, (kw,kh)) or
 * np.ones): umlClassdiagram.py:24-28; umlSequencediagram.py:27-31;
 * umlActivitydiagram.py:38-42; umlStateDiagram.py:24-28;
 * NewVersion2/img2UML.py:34-35; newVersion3/detect_arrow.py:24,28,39;
 * "New Version/detectArrow.py":10-11; NewVersion2/detectArrowv2.py:23-24.
 * anchor (-1,-1) = kernel centre.  `assume_binary` != 0 promises that every
 * pixel is 0 or one common non-zero value and selects the kernels that work on
 * one bit per pixel (for rectangles up to 33 x 31 after folding the iterations:
 * one pass, bytes in, bytes out).  0 = content unknown: the one-bit kernel
 * runs first (for the rectangles it covers), gives up on an image as soon as
 * it meets a third value, and the general 8-bit kernels redo only those
 * images.  Results do not depend on the choice when the promise holds. */
size_t ipa_morph_workspace_bytes(int width, int height, int batch);
int ipa_morph_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                 uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                 int width, int height, int batch,
                 int op, int kw, int kh, int anchor_x, int anchor_y, int iterations,
                 int assume_binary, void* ws, size_t ws_bytes, void* stream);

/* Device-side test used before choosing `assume_binary`: flags[b] (int32[batch])
 * becomes the common non-zero value of image b (0 for an all-zero image) or -1
 * when the image holds two different non-zero values. */
int ipa_binary_check_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                        int width, int height, int batch, int32_t* flags, void* stream);

/* ---- Canny ------------------------------------------------------------- */

/* cv2.Canny(gray, t1, t2, apertureSize=3, L2gradient=False): lineDetection.py:9;
 * "New Version/comparison.py":13; "Version4 with Chatgpt/contour_detection.py":61;
 * newVersion3/detectHorizontalVerticalLines.py:61; "New Version/detectArrow.py":8.
 * Hysteresis runs to its fixed point on the device, without any host interaction, for
 * every image size: one workgroup per image for images of at most 4096 tiles of 128x32
 * pixels (for batches of up to 48 images a first launch works on small regions side by
 * side and a second one finishes what crosses their borders); larger images are resolved
 * region by region in a fixed number of launches (run-time option "hyst_rounds"), after
 * which one workgroup per image follows whatever is left across region borders.  The call
 * never synchronises `stream` and copies nothing to the host. */
size_t ipa_canny_workspace_bytes(int width, int height, int batch);
int ipa_canny_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                 uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                 int width, int height, int batch,
                 double threshold1, double threshold2,
                 void* ws, size_t ws_bytes, void* stream);

/* ipa_canny_u8 for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  It has no counterpart
 * in the reference (cv2.Canny allocates its own buffers on every call).  The per-tile
 * scratch of the detector lives in a device array inside the library,
 * IPA_CANNY_STATIC_SCRATCH_BYTES per GPU, held until the process ends; the call makes no
 * hipMalloc / hipFree and takes no workspace.  Same kernels and results as ipa_canny_u8.
 * Batches that need more scratch than the array holds (ipa_canny_workspace_bytes() >
 * ipa_canny_static_max_bytes(): more than about 174 000 tiles of 128x32 pixels, e.g. 85
 * images of 3840x2160) are refused with IPA_ERR_WORKSPACE.  The device of `stream` must be
 * the current device (checked: IPA_ERR_ARG otherwise).  The call neither synchronises nor
 * copies to the host, whatever the image size, and is therefore safe under stream capture.
 * All calls on one GPU share the array: calls that overlap in time on
 * one GPU (different streams or threads) race and must be serialised by the caller. */
#define IPA_CANNY_STATIC_SCRATCH_BYTES ((size_t)512 * 1024)
size_t ipa_canny_static_max_bytes(void);
int ipa_canny_static_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                        uint8_t* dst, size_t dst_pitch, size_t dst_stride,
                        int width, int height, int batch,
                        double threshold1, double threshold2, void* stream);

/* ---- standard Hough transform ------------------------------------------ */

/* Accumulator geometry of cv2.HoughLines(img, rho, theta, threshold)
 * ("New Version/comparison.py":20,30): numangle rows of numrho bins, stored with
 * a one-bin border as int32[(numangle+2)*(numrho+2)]. */
int ipa_hough_dims(int width, int height, double rho, double theta, int* numangle, int* numrho);

/* Number of 512-column segments of one padded accumulator row (numrho+2 columns):
 * the unit of the optional `seg_max` arrays below. */
int ipa_hough_segments(int numrho);

/* Vote accumulator for every non-zero pixel.  `accum` = int32 device array,
 * image b at accum + b*accum_stride (elements); fully written by the call.
 * `seg_max` (optional, may be NULL) = int32[batch][numangle][ipa_hough_segments(numrho)]:
 * receives the largest count of every 512-column segment of every accumulator row
 * (an upper bound is also legal); ipa_hough_peaks_i32 uses it to skip segments that
 * cannot hold a peak.  It has no counterpart in cv2, which keeps the accumulator
 * private.
 * `max_points` bounds the per-image list of non-zero pixels kept in scratch
 * (0 = width*height); `point_counts` (int32[batch], device) receives the true
 * number of non-zero pixels -- if any exceeds max_points the accumulator of
 * that image is incomplete and the caller must retry with a larger bound. */
size_t ipa_hough_workspace_bytes(int width, int height, int batch, int max_points);
int ipa_hough_accum_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                       int width, int height, int batch, double rho, double theta,
                       int32_t* accum, size_t accum_stride, int32_t* seg_max,
                       int max_points, int32_t* point_counts,
                       void* ws, size_t ws_bytes, void* stream);

/* Local maxima of the accumulator above `threshold` (the findLocalMaximums step
 * of cv2.HoughLines).  For image b, keys + b*capacity receives up to `capacity`
 * 64-bit keys (votes << 32 | (0xFFFFFFFF - accumulator index)), in no particular
 * order; sorting them in DESCENDING order yields cv2's line order.
 * peak_counts[b] receives the number found (may exceed capacity: then only the
 * first `capacity` were kept and the caller retries).  `seg_max` = the array
 * ipa_hough_accum_u8 / ipa_front_end_u8 filled for these accumulators, or NULL
 * (then every cell is read). */
int ipa_hough_peaks_i32(const int32_t* accum, size_t accum_stride, const int32_t* seg_max,
                        int numangle, int numrho, int batch,
                        int threshold, uint64_t* keys, int capacity, int32_t* peak_counts, void* stream);

/* ---- segments: external contours -> bounding rectangles --------------------- */

/* get_segments(lines) of the parsers (umlClassdiagram.py:33-36, umlSequencediagram.py:36-39,
 * umlActivitydiagram.py:47-50, umlStateDiagram.py:33-36):
 *     contours, _ = cv2.findContours(lines, cv2.RETR_EXTERNAL, cv2.CHAIN_APPROX_SIMPLE)
 *     segments = [cv2.boundingRect(c) for c in contours]
 * For image b, boxes + b*max_boxes*4 receives up to max_boxes rectangles (x, y, w, h) of the
 * 8-connected components of the non-zero pixels that are not enclosed by another component,
 * in cv2's order (by first pixel in raster order, last first); counts[b] receives their number
 * (may exceed max_boxes: then the caller retries with more room), or -1 when the image has
 * more row segments (runs of equal pixels within a row) than `max_segments` (0 = a default of
 * about width*height/64): retry with a larger bound, at most width*height. */
size_t ipa_segments_workspace_bytes(int width, int height, int batch, int max_segments);
int ipa_segments_u8(const uint8_t* mask, size_t mask_pitch, size_t mask_stride,
                    int width, int height, int batch, int max_segments,
                    int32_t* boxes, int max_boxes, int32_t* counts,
                    void* ws, size_t ws_bytes, void* stream);

/* cv2.HoughLinesP(edges, rho, theta, threshold, minLineLength=, maxLineGap=)
 * (lineDetection.py:12-19, "New Version/comparison.py":14,
 * newVersion3/detectHorizontalVerticalLines.py:64): the progressive probabilistic Hough transform.
 * The visits of the non-zero pixels are sequential by definition (pseudo-random order from
 * cv::RNG, every visit changes accumulator and mask), so one workgroup of sixteen waves runs one
 * image (the waves share the bulk work inside a visit) and a batch runs as many workgroups side by side.  For image b, lines + b*max_lines*4 receives up to max_lines
 * segments (x0, y0, x1, y1) in the order cv2 finds them; counts[b] receives their number (may
 * exceed max_lines: retry with more room) or -1 when the image has more non-zero pixels than
 * `max_points` (0 = width*height, always enough).  The accumulator has numangle x numrho cells
 * (ipa_hough_lines_p_dims: numrho = round((2 (width + height) + 1) / rho)) per image in `ws`. */
int ipa_hough_lines_p_dims(int width, int height, double rho, double theta, int* numangle, int* numrho);
size_t ipa_hough_lines_p_workspace_bytes(int width, int height, int batch, double rho, double theta, int max_points);
int ipa_hough_lines_p_u8(const uint8_t* src, size_t src_pitch, size_t src_stride,
                         int width, int height, int batch, double rho, double theta,
                         int threshold, int min_line_length, int max_line_gap, int max_points,
                         int32_t* lines, int max_lines, int32_t* counts,
                         void* ws, size_t ws_bytes, void* stream);

/* ---- fused front-end ---------------------------------------------------- */

/* One pass of the whole raster front-end over a batch of BGR images: the
 * composition the reference's parsers run before their contour logic --
 * load_image_from_file / threshold_image / extract_lines
 * (umlClassdiagram.py:12-30, umlSequencediagram.py:15-33,
 * umlActivitydiagram.py:22-44, umlStateDiagram.py:12-30) plus the line-detection
 * prefix cvtColor -> [GaussianBlur 5x5] -> Canny -> HoughLines
 * (lineDetection.py:6-9, newVersion3/detectHorizontalVerticalLines.py:55-64,
 * "New Version/comparison.py":30):
 *   gray   = cvtColor(bgr, BGR2GRAY)
 *   thresh = threshold(gray, thresh_val, 255, THRESH_BINARY_INV)
 *   vlines = morphologyEx(thresh, OPEN, rect(1, vert_len), iterations)
 *   hlines = morphologyEx(thresh, OPEN, rect(horiz_len, 1), iterations)
 *   edges  = Canny(blur5x5 ? GaussianBlur(gray,(5,5),0) : gray, canny_lo, canny_hi)
 *   accum  = vote accumulator of HoughLines(edges, 1, pi/180, .)   (skipped if NULL;
 *            seg_max as for ipa_hough_accum_u8, may be NULL)
 * The five output planes share one pitch/stride.  edges = NULL skips Canny (the parsers
 * only need the first four planes); accum must then be NULL as well.  Never synchronises
 * `stream`.  Batches of about 1.5 gigapixels or more run as chunks
 * whose stages overlap on helper streams of the library (the vote of one chunk beside
 * the binarisation of the next, the opens beside Canny); all of it is joined back into
 * `stream` before the call returns, so the caller sees ordinary stream order. */
size_t ipa_front_end_workspace_bytes(int width, int height, int batch, int max_points);
/* Number of chunks ipa_front_end_u8 would cut this batch into with the current options (1 = one
 * chain; 0 for an empty batch).  Diagnostics: tests use it to know which schedule they exercised. */
int ipa_front_end_chunks(int width, int height, int batch);
int ipa_front_end_u8(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride,
                     uint8_t* gray, uint8_t* thresh, uint8_t* vlines, uint8_t* hlines, uint8_t* edges,
                     size_t plane_pitch, size_t plane_stride,
                     int32_t* accum, size_t accum_stride, int32_t* seg_max,
                     int max_points, int32_t* point_counts,
                     int width, int height, int batch,
                     int thresh_val, int vert_len, int horiz_len, int iterations,
                     double canny_lo, double canny_hi, int blur5x5,
                     void* ws, size_t ws_bytes, void* stream);


/* ipa_front_end_u8 that also leaves the two line masks as 1-bit planes (bit i of word j of a row = pixel
 * 32 j + i is set), the form get_segments works on: extract_lines followed by get_segments on both masks
 * (umlClassdiagram.py:27-36, umlSequencediagram.py:30-39, umlActivitydiagram.py:41-50,
 * umlStateDiagram.py:27-36) then needs no pass that packs the byte masks again.  Rows are
 * ipa_bits_pitch_words(width) words apart, images height rows apart; 16-byte aligned; either may be NULL. */
size_t ipa_bits_pitch_words(int width);
int ipa_front_end_bits_u8(const uint8_t* bgr, size_t bgr_pitch, size_t bgr_stride,
                          uint8_t* gray, uint8_t* thresh, uint8_t* vlines, uint8_t* hlines, uint8_t* edges,
                          size_t plane_pitch, size_t plane_stride,
                          int32_t* accum, size_t accum_stride, int32_t* seg_max,
                          int max_points, int32_t* point_counts,
                          int width, int height, int batch,
                          int thresh_val, int vert_len, int horiz_len, int iterations,
                          double canny_lo, double canny_hi, int blur5x5,
                          void* ws, size_t ws_bytes, void* stream,
                          uint32_t* vlines_bits, uint32_t* hlines_bits);

/* ipa_segments_u8 on a mask that already is a 1-bit plane (as written by ipa_front_end_bits_u8; bits beyond the
 * width must be clear): get_segments of the parsers, umlClassdiagram.py:33-36.  Same outputs and workspace. */
int ipa_segments_bits_u32(const uint32_t* bits, size_t bits_pitch_words, size_t bits_stride_words,
                          int width, int height, int batch, int max_segments,
                          int32_t* boxes, int max_boxes, int32_t* counts,
                          void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* IPAMD_H */
