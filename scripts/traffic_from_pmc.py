"""Turns the FETCH_SIZE / WRITE_SIZE passes of scripts/pmc_passes.sh into per-stage HBM traffic
per launch (bytes), written as profiles/traffic.json -> {"<size>x<batch>": {stage: bytes}}.

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts a wide coalesced
streaming read at half its bytes (128-B requests tallied as 64 B), so it is doubled here;
WRITE_SIZE is taken as is.  The SQ_INSTS_VALU pass (wave instructions) goes into "valu_per_image" the same way.
usage: traffic_from_pmc.py <pmc dir> <key e.g. 4kx128> [out json]"""
import collections, csv, glob, json, os, re, sys

STAGE_OF = {
    "k_bgr2gray": "gray_thresh_pack", "k_bgr2gray_flat": "gray_thresh_pack", "k_bitmorph_v": "open_vertical", "k_bitmorph_h3": "open_horizontal",
    "k_bitmorph": "open_horizontal", "k_canny_stream": "canny_nms", "k_canny_nms": "canny_nms",
    "k_hyst_tiles": "canny_hysteresis", "k_hyst_tiles_lds": "canny_hysteresis", "k_hyst_drain": "canny_hysteresis", "k_hyst_final": "canny_hysteresis", "k_hough_compact": "hough_compact",
    "k_hough_vote": "hough_vote", "k_hough_vote_s": "hough_vote", "k_hough_peaks": "hough_peaks",
    "k_hough_peaks_seg": "hough_peaks", "k_hough_compact2": "hough_compact", "k_gauss5_stream": "gaussian_blur",
    "k_canny_wide": "gray_thresh_canny",
}


def per_kernel(root, counter):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            k = re.sub(r"[<(].*", "", k)
            vals[k].append(float(r["Counter_Value"]))
    return vals


def main():
    root, key = sys.argv[1], sys.argv[2]
    out = sys.argv[3] if len(sys.argv) > 3 else os.path.join(os.path.dirname(__file__), "..", "profiles", "traffic.json")
    fetch = per_kernel(os.path.join(root, "fetch"), "FETCH_SIZE")
    write = per_kernel(os.path.join(root, "write"), "WRITE_SIZE")
    stage = collections.defaultdict(float)
    rows = []
    # passes of the front-end in the run = launches of the gray kernel (one per pass); a kernel that is launched
    # several times per pass (the hysteresis rounds of a huge image) counts with all of them
    def passes(vals):
        for g in ("k_bgr2gray_flat", "k_bgr2gray", "k_canny_wide"):
            if vals.get(g):
                return len(vals[g])
        return None
    for k in sorted(set(fetch) | set(write)):
        if k not in STAGE_OF:
            continue
        # (the first pass is the planning run; all passes do the same work)
        nf = min(passes(fetch) or 1 << 30, len(fetch.get(k) or [1]))  # (a kernel with fewer launches than passes: per launch)
        nw = min(passes(write) or 1 << 30, len(write.get(k) or [1]))
        f = 2.0 * 1024 * (sum(fetch[k]) / nf) if fetch.get(k) else 0.0
        w = 1024.0 * (sum(write[k]) / nw) if write.get(k) else 0.0
        stage[STAGE_OF[k]] += f + w
        rows.append((k, len(fetch.get(k, [])), f, w))
    data = {}
    if os.path.exists(out):
        data = json.load(open(out))
    data[key] = {s: round(v) for s, v in stage.items()}
    valu = collections.defaultdict(float)
    vals = per_kernel(os.path.join(root, "sq2"), "SQ_INSTS_VALU")
    for k, v in vals.items():
        if k in STAGE_OF and v:
            valu[STAGE_OF[k]] += sum(v) / min(passes(vals) or 1 << 30, len(v))
    m = re.fullmatch(r"(\w+?)x(\d+)", key)
    if m:  # bytes per image and stage: what bench.py scales by its own batch
        data.setdefault("per_image", {})[m.group(1)] = {s: v / int(m.group(2)) for s, v in stage.items()}
        if valu:  # VALU wave instructions per image and stage
            data.setdefault("valu_per_image", {})[m.group(1)] = {s: v / int(m.group(2)) for s, v in valu.items()}
    data.setdefault("_note", "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), rocprofv3 --pmc, separate passes")
    json.dump(data, open(out, "w"), indent=1, sort_keys=True)
    for k, n, f, w in rows:
        print(f"{k:20s} launches={n:4d} fetch={f/1e6:10.1f} MB  write={w/1e6:10.1f} MB")
    for s, v in sorted(valu.items()):
        print(f"valu {s:20s} {v/1e9:8.3f} G wave instructions per launch")


if __name__ == "__main__":
    main()
