#!/usr/bin/env python3
"""Headline benchmark: megapixels/second through the full raster front-end
(BGR -> gray -> binary threshold -> vertical/horizontal line opening -> Canny ->
Hough accumulator + peaks) on synthetic UML-like rasters.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--size 4k|1080p|16k] [--no-cpu]

N > 1 runs one process per GPU: either under ``python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N`` or, started plainly as ``python bench.py --gpus N``, by launching exactly that as a child process
and passing its output and exit code through.  By default the N > 1 workload is a FIXED total of 4096 4K diagrams (seeds
4000..8095) split by image index over the ranks (``sharding.shard_range``; 2048 / 1024 / 512 images
per GPU at N = 2 / 4 / 8), each rank running the whole front-end on its block; ``--weak`` gives every
rank 1024 images of its own instead.  No collective touches the data path: the only communication
is the barrier, the max-reduce of the elapsed time and, outside the timed region, the host gather of
per-image checksums to rank 0 (``sharding.gather_to_rank0``), which checks a sample of them against
its own single-GPU run of the same seeds.  N = 1 runs 1024 diagrams (seeds 2000..3023) on one GPU.

A step = one pass of ``ipa_front_end_u8`` + ``ipa_hough_peaks_i32`` over a batch that is
already resident in HBM.  The JSON line also carries the roofline of the dominant kernel
(HIP-event timed inside the timed steps) and the CPU baseline (the C restatement in
``oracle/`` on the host cores, bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SIZES = {"1080p": (1080, 1920), "4k": (2160, 3840), "16k": (16384, 16384)}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def _gen_one(args):
    from imageprocessing_amd.synth import synth_diagram

    shm_name, idx, seed, h, w, kw = args
    from multiprocessing import shared_memory

    shm = shared_memory.SharedMemory(name=shm_name)
    try:
        arr = np.ndarray((h, w, 3), dtype=np.uint8, buffer=shm.buf, offset=idx * h * w * 3)
        arr[...] = synth_diagram(seed, h, w, **kw)
    finally:
        shm.close()
    return idx


class HostGenerator:
    """Forked worker pool + one shared-memory chunk, created BEFORE the GPU is initialised
    (workers never touch the GPU).  Falls back to in-process generation."""

    def __init__(self, n, h, w, workers):
        self.n, self.h, self.w = n, h, w
        self.chunk = max(1, min(n, (1 << 30) // (h * w * 3)))
        self.pool = None
        self.shm = None
        if workers > 1 and n > 1:
            try:
                import multiprocessing as mp
                from multiprocessing import shared_memory

                self.shm = shared_memory.SharedMemory(create=True, size=self.chunk * h * w * 3)
                self.pool = mp.get_context("fork").Pool(workers)
            except Exception:
                self.close()

    def chunks(self, seed0, kw):
        from imageprocessing_amd.synth import synth_diagram

        n, h, w = self.n, self.h, self.w
        for c0 in range(0, n, self.chunk):
            m = min(self.chunk, n - c0)
            if self.pool is not None:
                list(self.pool.imap_unordered(_gen_one, [(self.shm.name, i, seed0 + c0 + i, h, w, kw) for i in range(m)]))
                host = np.ndarray((m, h, w * 3), dtype=np.uint8, buffer=self.shm.buf)
            else:
                host = np.stack([synth_diagram(seed0 + c0 + i, h, w, **kw) for i in range(m)]).reshape(m, h, w * 3)
            yield c0, m, host

    def close(self):
        if self.pool is not None:
            self.pool.close()
            self.pool.join()
            self.pool = None
        if self.shm is not None:
            self.shm.close()
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass
            self.shm = None


def generate_to_device(gen, seed0, kw):
    """Synthetic diagrams -> pitched device batch, uploaded chunk by chunk."""
    import torch
    from imageprocessing_amd._device import DevImage

    img = DevImage.empty(gen.n, gen.h, gen.w, 3)
    try:
        for c0, m, host in gen.chunks(seed0, kw):
            img.t[c0 : c0 + m, : gen.h, : gen.w * 3].copy_(torch.from_numpy(host))
            torch.cuda.synchronize()
    finally:
        gen.close()
    return img


def cpu_baseline(h, w, kw, seconds_target=12.0):
    """The oracle's front-end (C restatement, OpenMP) on the host cores over a bounded sample."""
    from imageprocessing_amd.synth import synth_diagram
    from oracle import frontref as fr

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    fr.lib()
    imgs = [synth_diagram(2000 + i, h, w, **kw) for i in range(2)]
    fr.front_end(imgs[0])  # warm
    t0 = time.perf_counter()
    n = 0
    while True:
        fr.front_end(imgs[n % len(imgs)])
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds_target or n >= 512:
            break
    mpix = n * h * w / 1e6 / el
    return {"value": round(mpix, 2), "unit": "Mpix/s", "cores": threads, "kind": "port",
            "sample": f"{n} x {w}x{h} synthetic diagrams, full front-end incl. Hough accumulator, "
                      f"oracle/frontref.c with OpenMP ({threads} threads), {el:.1f} s"}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children and wait for them."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default 1024 for 4k, 1 otherwise)")
    ap.add_argument("--size", default="4k", choices=list(SIZES))
    ap.add_argument("--hough-threshold", type=int, default=200)
    ap.add_argument("--blur", action="store_true", help="variant B: GaussianBlur 5x5 before Canny")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--check", type=int, default=2, help="images per rank verified against the oracle after the run")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="backend of the barrier / max-time reduction (gloo: rehearsal of the N>1 path without N GPUs)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--weak", action="store_true", help="N > 1: --batch images per rank instead of a fixed total split over the ranks")
    ap.add_argument("--total", type=int, default=4096, help="N > 1 without --weak: images in total (split by index over the ranks)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # Started by itself: run the N ranks as CHILD processes of this one (torch.distributed.run), pass their
            # output through and leave with their exit code.  Nothing here has touched the GPU yet - torch is not
            # even imported - and this process never becomes another program.
            sys.exit(launch_ranks(args.gpus))
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher must start one process per GPU",
              file=sys.stderr)
        sys.exit(2)
    h, w = SIZES[args.size]
    batch = args.batch or (1024 if args.size == "4k" else 1)
    kw = {"noise": 6, "gradient": True} if args.size == "16k" else {}
    line_len = 101 if args.size == "16k" else 15
    seed0 = {"1080p": 1000, "4k": 2000, "16k": 9000}[args.size] + rank * batch
    # N > 1: a fixed total split by image index over the ranks (strong scaling), unless --weak
    sharded = world > 1 and not args.weak and args.size == "4k" and not args.batch
    total = world * batch
    shard_lo = rank * batch
    if sharded:
        from imageprocessing_amd.sharding import shard_range

        total = args.total
        shard_lo, shard_hi = shard_range(total, rank, world)
        batch = shard_hi - shard_lo
        seed0 = 4000 + shard_lo

    workers = max(1, min(16, (os.cpu_count() or 1) // max(world, 1)))
    gen = HostGenerator(batch, h, w, workers)  # forks its workers before the GPU is touched

    import torch
    import torch.distributed as dist

    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import imageprocessing_amd as ipa
    from imageprocessing_amd import _native

    L = _native.lib()
    dev = torch.device("cuda", local_rank)
    t_gen = time.perf_counter()
    bgr = generate_to_device(gen, seed0, kw)
    t_gen = time.perf_counter() - t_gen

    fe = ipa.FrontEnd(batch, h, w, thresh_val=128, vert_len=line_len, horiz_len=line_len, iterations=2,
                      canny=(50, 150), blur=args.blur, hough=True)
    cap = 4096
    keys = torch.empty((batch, cap), dtype=torch.int64, device=dev)
    peak_counts = torch.zeros(batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        fe.enqueue(bgr)
        rc = L.ipa_hough_peaks_i32(fe.accum.data_ptr(), fe.accum.stride(0), fe.seg_max.data_ptr(), fe.na, fe.nr, batch,
                                   args.hough_threshold,
                                   keys.data_ptr(), cap, peak_counts.data_ptr(), stream)
        if rc:
            raise RuntimeError(_native.last_error())

    # first step also sizes the Hough point lists (re-plans on overflow)
    fe.run(bgr)
    for _ in range(max(args.warmup - 1, 0)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # Stage durations come from a separate pass outside the timed region: with timing enabled the
    # library runs its stages one after the other over the whole batch (in the timed steps above
    # it overlaps independent stages on helper streams, so their durations would not add up).
    prof_steps = min(args.steps, 3)
    L.ipa_profile_enable(1)
    for _ in range(prof_steps):
        step()
    torch.cuda.synchronize()
    L.ipa_profile_enable(0)
    nst = L.ipa_profile_stage_count()
    ms = (C.c_double * nst)()
    cnt = (C.c_int * nst)()
    L.ipa_profile_collect(ms, cnt)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-image checksums of every output (sums of the five planes, of the accumulator, and the number of edge
    # pixels), gathered on the host of rank 0 outside the timed region
    def checksums(fe_, n):
        rows = []
        for i0 in range(0, n, 16):
            i1 = min(i0 + 16, n)
            cols = [fe_.plane(k).t[i0:i1, :h, :w].sum(dim=(1, 2), dtype=torch.int64) for k in range(5)]
            cols.append(fe_.accum[i0:i1].sum(dim=(1, 2), dtype=torch.int64))
            cols.append(fe_.point_counts[i0:i1].to(torch.int64))
            rows.append(torch.stack(cols, dim=1))
        return torch.cat(rows).cpu().numpy() if rows else np.zeros((0, 7), np.int64)

    gather_note = None
    sums_bad = 0
    if world > 1 and sharded:
        from imageprocessing_amd.sharding import gather_to_rank0

        mine = checksums(fe, batch)
        cpu_side = args.dist_backend != "nccl"
        all_sums, err = None, None
        try:
            host_group = dist.new_group(backend="gloo") if not cpu_side else None
            all_sums = gather_to_rank0(mine, total, rank, world, group=host_group)
        except Exception as e:  # decided below by ALL ranks together: a rank must never switch collectives alone
            err = e
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int64, device="cpu" if cpu_side else dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            how = "host gather (gloo, point to point)"
        else:
            # no usable host transport on at least one rank: every rank sends its rows through one all_gather of
            # the main group instead
            sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
            pad = torch.zeros((max(sizes), 7), dtype=torch.int64, device="cpu" if cpu_side else dev)
            pad[:batch] = torch.from_numpy(mine).to(pad.device)
            out_t = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(out_t, pad)
            all_sums = np.concatenate([o[:n].cpu().numpy() for o, n in zip(out_t, sizes)]) if rank == 0 else None
            how = f"all_gather of the main group (host gather failed on a rank: {type(err).__name__ if err else 'elsewhere'})"
        if rank == 0:
            # a sample of the other ranks' images, recomputed here on this GPU from the same seeds
            from imageprocessing_amd.synth import synth_diagram

            assert all_sums.shape == (total, 7)
            ids = sorted(set(np.linspace(0, total - 1, 2 * world + 1).astype(int).tolist()))
            one = ipa.FrontEnd(1, h, w, thresh_val=128, vert_len=line_len, horiz_len=line_len, iterations=2,
                               canny=(50, 150), blur=args.blur, hough=True)
            from imageprocessing_amd._device import upload

            for i in ids:
                one.run(upload(synth_diagram(4000 + i, h, w, **kw), 3))
                sums_bad += int(not np.array_equal(checksums(one, 1)[0], all_sums[i]))
            gather_note = (f"{total} per-image checksum rows on rank 0 by {how}; {len(ids)} images of all ranks recomputed "
                           f"on rank 0: {'all equal' if sums_bad == 0 else str(sums_bad) + ' DIFFER'}")

    # correctness spot check outside the timed region (oracle = checker only)
    check_note = "skipped"
    bad = 0
    if args.check > 0:
        from oracle import frontref as fr

        ids = sorted(set(np.linspace(0, batch - 1, min(args.check, batch)).astype(int).tolist()))
        for i in ids:
            src = bgr.t[i, :h, : w * 3].cpu().numpy().reshape(h, w, 3)
            ref = fr.front_end(src, 128, line_len, 2, 50, 150)
            if args.blur:
                g = fr.bgr2gray(src)
                e = fr.canny(fr.gaussian_blur(g, (5, 5), 0), 50, 150)
                ref = (ref[0], ref[1], ref[2], ref[3], e, fr.hough_accum(e, 1.0, np.pi / 180))
            for k in range(5):
                got = fe.plane(k).t[i, :h, :w].cpu().numpy()
                bad += int(not np.array_equal(got, ref[k]))
            bad += int(not np.array_equal(fe.accum[i].cpu().numpy(), ref[5]))
        check_note = f"{len(ids)} images x 6 outputs bit-exact vs oracle" if bad == 0 else f"{bad} MISMATCHING outputs"
        if bad:
            print(f"bench.py: rank {rank}: parity check failed: {check_note}", file=sys.stderr)
    bad += sums_bad
    bad_all = bad
    if world > 1:  # rank 0 reports the checks of every rank
        tb = torch.tensor([bad], dtype=torch.int64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(tb)
        bad_all = int(tb.item())

    if rank == 0:
        npix = float(batch) * h * w
        acc_bytes = 4.0 * (fe.na + 2) * (fe.nr + 2)
        stage_names = [L.ipa_profile_stage_name(i).decode() for i in range(nst)]
        # algorithmic bytes each stage has to move per image: its compulsory inputs and outputs,
        # counted once (intermediate re-reads earn nothing)
        alg = {
            "gray_thresh_pack": 5.0 * h * w,          # BGR in (3) + gray out (1) + thresh out (1)
            "open_vertical": 1.125 * h * w,           # 1 bit/px in + 1 B/px out
            "open_horizontal": 1.125 * h * w,
            "gaussian_blur": 2.0 * h * w,
            "canny_nms": 2.0 * h * w,                 # gray in + edges out
            "gray_thresh_canny": 6.0 * h * w,         # BGR in (3) + gray, thresh and edge codes out (3): the fused kernel
            "canny_hysteresis": 0.0,
            "hough_compact": 1.0 * h * w,             # edges in (the point list is << 1 B/px)
            "hough_vote": acc_bytes,                  # accumulator written once
            "hough_peaks": acc_bytes,                 # accumulator read once
        }
        stages = {}
        for i, nm in enumerate(stage_names):
            if cnt[i]:
                # one launch per stage and step in the (serial) profiling pass
                per = ms[i] / prof_steps
                # hysteresis and the guided peak search touch a data-dependent few percent of their
                # planes: no algorithmic-bytes figure is quoted for them
                quoted = per > 0 and nm not in ("canny_hysteresis", "hough_peaks")
                stages[nm] = {"ms": round(per, 4), "GBps": round(alg[nm] * batch / (per * 1e-3) / 1e9, 1) if quoted else None}
        dom = max(stages, key=lambda k: stages[k]["ms"]) if stages else None
        roof = None
        if dom:
            a = alg[dom] * batch / (stages[dom]["ms"] * 1e-3) / 1e9
            traffic = None
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tf):
                try:
                    # measured at 1024 images per launch; the kernels' traffic is proportional to the images
                    tj = json.load(open(tf))
                    per_image = tj.get("per_image", {}).get(args.size, {}).get(dom)
                    traffic = round(per_image * batch) if per_image is not None else None
                except Exception:
                    traffic = None
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(a / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "launch_ms": stages[dom]["ms"], "algorithmic_bytes_per_launch": alg[dom] * batch}
            # The whole step against its two floors, from the counters of the serial schedule (per image and stage,
            # scaled by the batch): every VALU wave instruction holds its SIMD for at least 4 clocks, every HBM
            # byte costs 1/peak.  Stages overlap in the timed steps, so both floors bound the step, not their sum.
            try:
                props = torch.cuda.get_device_properties(dev)
                vj = tj.get("valu_per_image", {}).get(args.size, {})
                pj = tj.get("per_image", {}).get(args.size, {})
                ran = [nm for nm in stages if nm in vj]
                if ran:
                    instr = sum(vj[nm] for nm in ran) * batch
                    simds = 4 * props.multi_processor_count
                    clock_hz = 2.4e9  # engine clock of the MI355X
                    floor_ms = instr * 4.0 / (simds * clock_hz) * 1e3
                    step_ms = elapsed / args.steps * 1e3
                    hbm = sum(pj.get(nm, 0.0) for nm in stages) * batch
                    roof["valu"] = {"wave_instructions_per_step": round(instr), "issue_floor_ms": round(floor_ms, 3),
                                    "frac_of_step": round(floor_ms / step_ms, 4),
                                    "per_stage_G": {nm: round(vj[nm] * batch / 1e9, 3) for nm in ran},
                                    "hbm_traffic_bytes_per_step": round(hbm),
                                    "hbm_floor_ms": round(hbm / (HBM_PEAK_GBS * 1e9) * 1e3, 3),
                                    "note": "SQ_INSTS_VALU and 2 x FETCH_SIZE + WRITE_SIZE per stage (rocprofv3 --pmc, serial "
                                            "schedule, profiles/traffic.json); floor = 4 clk per wave instruction on "
                                            f"{simds} SIMDs at 2.4 GHz"}
            except Exception:
                pass
            if "hough_vote" in stages and stages["hough_vote"]["ms"] > 0 and fe.hough:
                # The vote is bound by LDS atomics, not by HBM: also quote it against the LDS atomic rates measured on
                # this chip with 16 waves per CU (scripts/micro/lds_rate.hip, lds_rand.hip; clocks at 2.4 GHz):
                # a ds_add_u32 wave instruction takes 4.37 clk when its 64 addresses fall into different banks
                # (35.1e9 lane-atomics per second and CU) and 8.0-8.5 clk on addresses that are random and fresh for
                # every instruction, which is what Hough bins of scattered edge pixels are (18.7e9).
                props = torch.cuda.get_device_properties(dev)
                votes = float(fe.point_counts[:batch].sum().item()) * fe.na
                rate = votes / (stages["hough_vote"]["ms"] * 1e-3)
                free_rate = 35.1e9 * props.multi_processor_count
                rand_rate = 18.7e9 * props.multi_processor_count
                # (inside `roofline` while the vote is the dominant kernel, beside it otherwise)
                vote_lds = {"votes_per_s": round(rate, 0), "ceiling_votes_per_s": round(free_rate, 0),
                                       "frac": round(rate / free_rate, 4),
                                       "random_address_votes_per_s": round(rand_rate, 0),
                                       "frac_of_random_address_rate": round(rate / rand_rate, 4),
                                       "note": "180 votes per edge pixel, one ds_add_u32 lane each; ceiling = conflict-free rate; "
                                               "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.61 (profiles/pmc_r02j_4k_x1024_summary.txt; 0.68 before the comb mapping of the lanes)"}
                if dom == "hough_vote":
                    roof["lds_atomics"] = vote_lds
                else:
                    a2 = alg["hough_vote"] * batch / (stages["hough_vote"]["ms"] * 1e-3) / 1e9
                    roof["runner_up"] = {"kernel": "hough_vote", "bound": "hbm", "achieved": round(a2, 1), "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": round(a2 / HBM_PEAK_GBS, 4), "launch_ms": stages["hough_vote"]["ms"],
                                         "lds_atomics": vote_lds}
        cpu = None
        if not args.no_cpu:
            try:
                cpu = cpu_baseline(h, w, kw)
            except Exception as e:  # the baseline must never take the GPU number down with it
                cpu = {"error": str(e)}
        value = float(total) * h * w * args.steps / elapsed / 1e6
        whole = (8.0 * npix + acc_bytes * batch) * args.steps / elapsed / 1e9
        out = {
            "metric": "megapixels/sec through binarize+morph+Canny+Hough",
            "value": round(value, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": (f"{total} x {w}x{h} synthetic UML diagrams in total, split by image index over {world} GPUs "
                                    f"({batch} on rank 0), full front-end " if sharded else
                                    f"{batch} x {w}x{h} synthetic UML diagrams per GPU, full front-end ") +
                                   f"(gray, threshold 128 INV, open 1x{line_len}/{line_len}x1 it=2, "
                                   f"{'GaussianBlur 5x5, ' if args.blur else ''}Canny 50/150, Hough 1/pi/180 + peaks>"
                                   f"{args.hough_threshold})",
                       "images_per_gpu": batch, "images_total": total, "width": w, "height": h,
                       "sharding": "per-image, no collective"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "pipeline_algorithmic_GBps_per_gpu": round(whole, 1),
            "pipeline_roofline_frac": round(whole / HBM_PEAK_GBS, 4),
            "stages": stages,
            "stage_timing": f"HIP events, separate pass of {prof_steps} steps with the stages serialised (the timed steps overlap stages)",
            "parity_check": check_note,
            "parity_ok": (bad_all == 0) if (args.check > 0 or sharded) else None,
            "shard_gather": gather_note,
            "gen_seconds": round(t_gen, 1),
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    if bad_all:
        sys.exit(3)  # a throughput line next to wrong results must not read as a pass


if __name__ == "__main__":
    main()
