#!/usr/bin/env python3
"""Headline benchmark: clips/sec of the FULL perturbation-mask search on I3D.

A "step" = one pass of the hot path over one batch of synthetic clips per GPU:
baseline forward -> init_mask('central') -> 300 Adam iterations (freeze scan,
I3D forward, backward-data, reverse scan, TV/L1, Adam) -> reverse score ->
Grad-CAM, for `--batch` 16-frame 224x224 clips (BASELINE.json configs[1];
reference loop: FindMasksComparison_I3D_smth.py:166-277).  Clips are resident in
HBM before the timed region.  N>1: one process per GPU (torchrun), clips sharded
clip_id % world, no data-path collective; one RCCL all_gather of the fixed-size
per-clip records inside the timed region.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N
ranks itself (child `python -m torch.distributed.run ...`, before this process
touches the GPU) and exits with the child's code.

The headline arithmetic is `bf16x6`: every fp32 operand split into three bf16 planes (all 24
significand bits), six MFMA passes per k-step, fp32 accumulate, fp32 storage -- the reference's fp32
arithmetic (logits within 1e-6 of the exact-fp32 MFMA chain) on the bf16 matrix cores.

Prints ONE JSON line on rank 0.  At N=1 the same line also carries
  * `secondary.s32` / `secondary.s32_bf16act`: BASELINE configs[4], 32-frame clips, in the headline
    arithmetic and in its stated form (bf16 activation storage, 2 MFMA passes);
  * `secondary.bf16x3`: the 16-bit-significand split (3 passes; within north_star's 1e-3, not fp32);
  * `secondary.fp32_mfma`: the native fp32 MFMA (v_mfma_f32_32x32x2_f32) for reference;
  * `secondary.convlstm`: BASELINE configs[3], CLSTM_4 on [B,1,32,120,160], N=100;
  * `cpu_baseline`: the CPU oracle on the host cores (port), plus the
    reference-literal cost line (B=16 forward per iteration + weight gradients);
  * `summary`: the headline figures of every block again, compact, as the LAST key of the line.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks, HBM3E peak
PEAK_FP32_MFMA, PEAK_BF16_MFMA = 157.3, 2500.0
PEAK_HBM_GBS = 8000.0
# mode -> (dtype string of the JSON line, dense MFMA peak for THAT dtype, MFMA passes per algorithmic FLOP)
MODES = {
    "fp32": ("f32", PEAK_FP32_MFMA, 1, "fp32 MFMA (v_mfma_f32_32x32x2_f32), exact fp32 FMA chains"),
    "bf16x6": ("f32 (each operand as 3 bf16 planes = 24 significand bits, 6 MFMA passes, f32 accumulate, f32 storage)",
               PEAK_FP32_MFMA, 6,
               "results are fp32-class (logits within 1e-6 of the fp32 MFMA chain), so `peak` is the dense fp32 MFMA "
               "peak; the instructions are v_mfma_f32_32x32x16_bf16, 6 MFMA FLOPs per algorithmic FLOP: the ceiling of "
               "this form is 2500/6 = 417 TFLOP/s, see `matrix_pipe`"),
    "bf16x3": ("bf16x3 (each operand as 2 bf16 planes = 16 significand bits, 3 MFMA passes, f32 accumulate, f32 storage)",
               PEAK_BF16_MFMA, 3,
               "bf16 dense MFMA; every algorithmic FLOP costs 3 MFMA FLOPs, so the ceiling of this form is 833 TFLOP/s"),
    "bf16act": ("bf16 activations/gradients in HBM (RNE), weights as 2 bf16 planes, 2 MFMA passes, f32 accumulate",
                PEAK_BF16_MFMA, 2,
                "bf16 dense MFMA; every algorithmic FLOP costs 2 MFMA FLOPs, so the ceiling of this form is 1250 TFLOP/s"),
}
HEADLINE_MATH = "bf16x6"
NCLASS = 96   # IVF_PROFILE_CLASSES
GATHER_CAMS = os.environ.get("IVF_BENCH_GATHER_CAMS") == "1"   # also all-gather the Grad-CAM maps (SURVEY 8e's optional payload)
PMC_FILE = os.path.join(ROOT, "profiles", "r03_pmc_hbm_traffic.json")


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="clips per GPU per step (searched together)")
    ap.add_argument("--iters", type=int, default=300, help="Adam iterations per search (reference N=300)")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary blocks (other arithmetic modes, S32, ConvLSTM)")
    ap.add_argument("--cpu-sample-iters", type=int, default=10)
    ap.add_argument("--tuning", default=None,
                    help="JSON file with a per-layer kernel choice: loaded instead of autotuning when it exists and fits "
                         "this plan, written after autotuning otherwise (profiling runs: no tuner launches in the trace)")
    ap.add_argument("--math", choices=sorted(MODES), default=HEADLINE_MATH,
                    help="arithmetic of the Unit3D convolutions of the headline block")
    return ap.parse_args()


def self_launch(args):
    """Plain `python bench.py --gpus N`: start N ranks as a CHILD torchrun (never exec: this
    process must not be replaced) and relay its output and exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rc = subprocess.run(cmd, env=env).returncode
    raise SystemExit(rc)


def host_cores():
    """CPU share of this process: cgroup quota if set, else the affinity mask."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(n_iter_total, sample_iters, lam1, lam2, threads, budget_s=40.0):
    """The CPU oracle (torch fp32 restatement of the reference path, oracle/) on the host cores of
    this box.  Port leg: 2 clips, each init_mask + `sample_iters` iterations + reverse score +
    Grad-CAM, the iteration cost extrapolated linearly to n_iter_total.  Reference-literal leg: what
    the published loop costs per iteration as written (smth:202-205,213: the WHOLE batch of 16
    forwarded with one shared mask, backward with weight gradients) -- reported beside the port so the
    16x algorithmic saving of SURVEY F10/F11 is not read as hardware speed-up."""
    import torch
    import ivf_recipe as R
    from oracle import gradcam_ref, i3d_ref, mask_ref
    torch.set_num_threads(threads)
    sd = R.to_torch(R.i3d_state_dict(num_classes=174))
    t_start = time.perf_counter()
    per_clip, notes = [], []
    for cid in (1000, 1001):
        x = torch.from_numpy(R.clip(cid % 16))[None]
        with torch.no_grad():
            out = i3d_ref.forward(x, sd)
        target = int(out[0].argmax())

        def score_fn(v):
            return i3d_ref.forward(v, sd)[0, target]
        t0 = time.perf_counter()
        with torch.no_grad():
            i3d_ref.forward(x, sd)
            tm, _ = mask_ref.init_mask_central(x, score_fn, 0.9, 'freeze')
        t_init = time.perf_counter() - t0
        n_s = sample_iters if t_init < 6 else max(2, sample_iters // 4)     # slow host: keep the leg bounded
        t0 = time.perf_counter()
        res = mask_ref.search_clip(x, score_fn, lam1, lam2, n_s, init=tm)   # includes the reverse score
        t_loop = time.perf_counter() - t0
        t0 = time.perf_counter()
        with torch.no_grad():
            score_fn(mask_ref.perturb_sequence(x, res['mask'], 'reverse'))
        t_rev = time.perf_counter() - t0
        t0 = time.perf_counter()
        gradcam_ref.gradcam_i3d(x, sd, None)
        t_gc = time.perf_counter() - t0
        per_iter = (t_loop - t_rev) / n_s
        per_clip.append(t_init + per_iter * n_iter_total + t_rev + t_gc)
        notes.append(f"init_mask {t_init:.1f}s + {n_s} of {n_iter_total} iterations ({per_iter:.3f}s each, "
                     f"extrapolated linearly) + reverse {t_rev:.2f}s + Grad-CAM {t_gc:.2f}s")
        if time.perf_counter() - t_start > budget_s:
            break
    total = sum(per_clip) / len(per_clip)
    out = dict(value=1.0 / total, unit="clips/s", cores=threads, kind="port",
               sample=f"{len(per_clip)} clip(s) [1,3,16,224,224], B=1 per-clip search: " + " | ".join(notes))
    # reference-literal iteration: batch of 16, shared mask, weight gradients on
    try:
        B = 16
        xb = torch.stack([torch.from_numpy(R.clip(c)) for c in range(B)])
        sdg = {k: (v.clone().requires_grad_() if v.is_floating_point() and 'running' not in k else v)
               for k, v in sd.items()}
        tm = torch.zeros(16).requires_grad_()
        t0 = time.perf_counter()
        mc = torch.sigmoid(tm)
        loss = lam1 * mc.abs().sum() + lam2 * mask_ref.calc_tv_norm(mc, 3, 3) \
            + i3d_ref.forward(mask_ref.perturb_sequence(xb, mc, 'freeze'), sdg)[0, 0]
        loss.backward()
        t_lit = time.perf_counter() - t0
        out["reference_literal"] = dict(
            value=1.0 / (t_lit * n_iter_total), unit="clips/s", cores=threads,
            sample=f"1 iteration as published (smth:202-213): batch of {B} clips forwarded with one shared mask, "
                   f"one row read, backward incl. weight gradients = {t_lit:.1f}s; x{n_iter_total} iterations per clip "
                   f"(init_mask / reverse / Grad-CAM not included)")
    except Exception as e:      # a host without the memory for the B=16 graph: report, do not fail the bench
        out["reference_literal"] = dict(value=None, note=f"not measured: {type(e).__name__}: {e}")
    return out


def collect_roofline(L, eng, B):
    """HIP-event sample of the conv launches of the timed region -> roofline object for the dominant LAUNCH
    SITE (one convolution of the plan in one direction: a kernel template may serve several layers, so the
    site is what has a definite algorithmic size)."""
    import numpy as np
    lib = L.lib()
    ns = lib.ivf_i3d_num_sites(eng._h)
    s_ms = (ctypes.c_double * ns)()
    s_n = (ctypes.c_longlong * ns)()
    s_fl = (ctypes.c_double * ns)()
    s_var = (ctypes.c_int * ns)()
    L.check(lib.ivf_profile_collect_sites(s_ms, s_n, s_fl, s_var, ns))
    ms = (ctypes.c_double * NCLASS)()
    launches = (ctypes.c_longlong * NCLASS)()
    flops = (ctypes.c_double * NCLASS)()
    L.check(lib.ivf_profile_collect(ms, launches, flops))
    L.check(lib.ivf_profile_disable())
    if sum(launches) == 0:
        return None
    name = lambda v: lib.ivf_profile_class_name(v).decode() or f"class{v}"
    math = eng.math
    _, peak, passes, peak_note = MODES[math]
    dom = int(np.argmax([s_ms[i] for i in range(ns)]))
    buf = ctypes.create_string_buffer(64)
    L.check(lib.ivf_i3d_site_name(eng._h, dom, buf))
    site = buf.value.decode()
    v = int(s_var[dom])
    avg_ms = s_ms[dom] / s_n[dom]
    gflop = s_fl[dom] / s_n[dom] / 1e9
    achieved = gflop / avg_ms          # GFLOP / ms = TFLOP/s
    tot_ms = sum(ms[i] for i in range(NCLASS))
    tot_fl = sum(flops[i] for i in range(NCLASS))
    shares = {name(i): round(ms[i] / tot_ms, 3) for i in range(NCLASS) if launches[i] > 0}
    # HBM bytes per launch of THAT site from this round's PMC passes (profiles/), taken on the same batch
    # size and the same kernel template; otherwise null
    # keyed by launch site, arithmetic mode, batch and frames; the kernel template the PMC run had tuned for that site is
    # quoted beside it (the tuner may pick another tile on another box: the compulsory bytes of the site do not change)
    traffic, src = None, None
    try:
        pm = json.load(open(PMC_FILE))
        ent = pm["sites"].get(site)
        if ent and pm.get("batch") == B and pm.get("math") == math and pm.get("frames", 16) == eng.clip_shape[1]:
            traffic = ent["hbm_bytes_per_launch"]
            src = (f"profiles/{os.path.basename(PMC_FILE)} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH "
                   f"doubled per the gfx950 rule; same launch site, arithmetic and batch as this run; kernel there: "
                   f"{ent['kernel']})")
    except (OSError, ValueError, KeyError):
        pass
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": src,
        "kernel": name(v), "launch_site": site, "avg_launch_ms": round(avg_ms, 4),
        "sampled_launches": int(s_n[dom]), "algorithmic_gflop_per_launch": round(gflop, 3),
        # rocprofv3 --stats averages per kernel NAME, i.e. over every site the template serves: the figure to
        # compare with profiles/r02_bench_kernel_stats.csv
        "kernel_name_avg_launch_ms": round(ms[v] / launches[v], 4), "kernel_name_sampled_launches": int(launches[v]),
        "mfma_passes_per_algorithmic_flop": passes,
        # what the matrix pipe itself is doing: MFMA FLOPs issued (algorithmic x passes) against the dense peak of the
        # MFMA instruction the kernel issues
        "matrix_pipe": ({"instruction": "v_mfma_f32_32x32x16_bf16", "peak": PEAK_BF16_MFMA,
                         "frac": round(achieved * passes / PEAK_BF16_MFMA, 4)} if math != "fp32" else
                        {"instruction": "v_mfma_f32_32x32x2_f32", "peak": PEAK_FP32_MFMA,
                         "frac": round(achieved / PEAK_FP32_MFMA, 4)}),
        "all_conv_kernels": {"achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                             "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4),
                             "share_of_sampled_conv_time": shares},
        "peak_dtype": peak_note,
    }


def timed_i3d(torch, dist, dev, rank, world, math, B, T, iters, steps, warmup, lam1, lam2, tuning_file=None):
    """W untimed + K timed steps of the full I3D search.  Returns (elapsed_s, roofline, engine math)."""
    import ivf_engine
    import ivf_lib as L
    import ivf_recipe as R
    import ivf_search
    import ivf_shard
    eng = ivf_engine.I3DEngine(174, (3, T, 224, 224), max_batch=B, softmax=True,
                               stride_mod_layers="" if T == 16 else "none", device=dev, math=math)
    saved = None
    if tuning_file and os.path.exists(tuning_file):
        try:
            doc = json.load(open(tuning_file))
            # (lib_version: the variant ids of a saved choice are only meaningful for the variant table that made them)
            if doc.get("batch") == B and doc.get("math") == eng.math and doc.get("frames") == T and \
                    doc.get("lib_version") == L.lib().ivf_version():
                saved = doc["variants"]
        except (OSError, ValueError, KeyError):
            saved = None
    eng.load_state_dict(R.i3d_state_dict(num_classes=174), autotune=(rank == 0 and saved is None))
    if saved is not None:
        try:
            eng.set_tuning(saved)
        except L.IvfError:
            eng.autotune()
            saved = None
    if tuning_file and saved is None and rank == 0:
        try:
            json.dump({"batch": B, "math": eng.math, "frames": T, "lib_version": L.lib().ivf_version(),
                       "variants": eng.get_tuning()}, open(tuning_file, "w"))
        except OSError:
            pass
    if dist is not None:
        # every rank must run the SAME kernel variant per layer (bit-identical per-clip results):
        # rank 0 tunes, one small broadcast installs its choice everywhere
        tune = torch.tensor(eng.get_tuning(), dtype=torch.int32, device=dev)
        dist.broadcast(tune, src=0)
        eng.set_tuning(tune.cpu().tolist())
    searcher = ivf_search.MaskSearch(eng, lam1, lam2, iters, "freeze", grad_cam_type="guessed", do_gradcam=True)
    # synthetic clips, resident in HBM before the timed region; shard: clip_id % world == rank
    total = warmup + steps
    clip_ids = [[(s * B + i) * world + rank for i in range(B)] for s in range(total)]
    uniq = sorted({c % 16 for ids in clip_ids for c in ids})      # 16 distinct synthetic clips, reused
    bank = {c: torch.from_numpy(R.clip(c, 3, T, 224, 224)).to(dev) for c in uniq}
    batches = [torch.stack([bank[c % 16] for c in ids]) for ids in clip_ids]
    labels = [[R.label(c, 174) for c in ids] for ids in clip_ids]

    def step(i):
        res = searcher.run(batches[i], labels[i])
        rec = ivf_search.pack_records(clip_ids[i], res, T, with_cam=GATHER_CAMS)   # (maps: 3.2 MB per clip, optional)
        return ivf_shard.gather_records(rec, equal_shards=True)    # ONE RCCL all_gather when world > 1

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    # sample conv launches on every 16th iteration with HIP events on the launch stream
    L.check(L.lib().ivf_profile_enable(16, 200000))
    fence()
    t0 = time.perf_counter()
    for i in range(warmup, total):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    roof = collect_roofline(L, eng, B) if rank == 0 else None
    if rank != 0:
        L.check(L.lib().ivf_profile_disable())
    return elapsed, roof, eng.math


def convlstm_block(torch, dev, B=256, C=1, iters=100, steps=2, warmup=1):
    """BASELINE configs[3]: CLSTM_4 (hidden 4, 2 layers, k5, stride 2, add_softmax) mask search on
    [B,1,32,120,160], N=100, lam 0.02/0.04 (KTH:105-118).  HBM/latency-bound: the roofline figure is
    algorithmic bytes per clip-iteration / time against the HBM peak."""
    import ivf_engine
    import ivf_recipe as R
    import ivf_search
    T, H, W, hid = 32, 120, 160, 4
    eng = ivf_engine.CLSTMEngine(6, (C, T, H, W), max_batch=B, hidden=hid, layers=2, kernel=5, stride=2,
                                 softmax=True, device=dev)
    eng.load_state_dict(R.clstm_state_dict(channels=C, tag=f'clstm{C}'))
    bank = [torch.from_numpy(R.clip(i, C, T, H, W) / 255.0).float().to(dev) for i in range(8)]
    x = torch.stack([bank[i % 8] for i in range(B)])
    s = ivf_search.MaskSearch(eng, 0.02, 0.04, iters, "freeze", do_gradcam=False)
    for _ in range(warmup):
        s.run(x, [0] * B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        s.run(x, [0] * B)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    clip_b = C * T * H * W * 4                                    # one clip-sized tensor (2.46 MB at C=1)
    gates = 4 * hid * (H // 2) * (W // 2) * 4 * T                 # first-layer gate planes i,f,g,o of all steps (9.83 MB)
    # freeze fwd (read X, write P) + net fwd (read P, write gates) + BPTT (read gates, write dP)
    #           + freeze bwd (read X, dP)
    per_clip_iter = 7 * clip_b + 2 * gates
    n_ci = B * steps * iters
    gbs = per_clip_iter * n_ci / dt / 1e9
    return {
        "metric": "clips/sec full mask-search (CLSTM_4, 32f, 100 iters)", "value": round(B * steps / dt, 2),
        "unit": "clips/s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 2),
        "dtype": "f32",
        "config": {"workload": f"ConvLSTM perturbation mask search, {iters} iters, synthetic KTH-shaped clips "
                               f"[{B},{C},32,120,160] (BASELINE configs[3]); init_mask + search + reverse score",
                   "clips_per_step": B, "iters": iters, "lam1": 0.02, "lam2": 0.04},
        "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
                     "scope": "whole search iteration (64 dependent cell steps each way: latency-bound)",
                     "algorithmic_bytes_per_clip_iteration": per_clip_iter,
                     "ms_per_iteration": round(dt / (steps * iters) * 1e3, 4)},
    }


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            self_launch(args)                     # before anything touches the GPU
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}: launch as `python bench.py "
                         f"--gpus N` (starts its own ranks) or with torch.distributed.run --nproc-per-node N")
    import torch
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # IVF_BENCH_FORCE_DIST=1: run the RCCL path (init, tuning broadcast, record all_gather, MAX all_reduce) with
    # a world of one -- how the N>1 code is exercised on a one-GPU box
    use_dist = world > 1 or os.environ.get("IVF_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL on ROCm

    lam1, lam2 = 0.01, 0.02   # FindMasksComparison_I3D_smth.py:106-113
    T, B = args.frames, args.batch
    cfg_name = {16: "BASELINE configs[1]", 32: "BASELINE configs[4] geometry"}.get(T, "")

    def i3d_block(math, frames, batch, steps, warmup, what, tuning=None, dist_=None, rank_=0, world_=1):
        """One timed block of the full I3D search -> the bench line's fields for it."""
        el, roof, m = timed_i3d(torch, dist_, dev, rank_, world_, math, batch, frames, args.iters, steps, warmup, lam1,
                                lam2, tuning_file=tuning)
        return {
            "metric": f"clips/sec full mask-search (I3D, {frames}f, {args.iters} iters)",
            "value": round(world_ * batch * steps / el, 4), "unit": "clips/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(el / steps * 1e3, 2), "dtype": MODES[m][0],
            "config": {"workload": what, "clips_per_gpu_per_step": batch, "iters": args.iters, "frames": frames,
                       "math": m},
            "roofline": roof}

    blk = i3d_block(args.math, T, B, args.steps, args.warmup,
                    f"I3D perturbation mask search, {args.iters} iters, synthetic clips [{B},3,{T},224,224] per GPU per step "
                    f"({cfg_name}); init_mask + search + reverse score + Grad-CAM",
                    tuning=args.tuning, dist_=dist, rank_=rank, world_=world)
    if rank == 0:
        out = {
            "metric": "clips/sec full mask-search (I3D, 16f, 300 iters)" if T == 16 else blk["metric"],
            "value": blk["value"], "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": blk["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": blk["dtype"], "data": "synthetic",
            "config": dict(blk["config"], lam1=lam1, lam2=lam2, sharding=f"clip_id % {world}, one all_gather of records"),
            "roofline": blk["roofline"],
        }
        summary = {"headline": {"math": args.math, "clips_per_s": blk["value"],
                                "roofline_frac": (blk["roofline"] or {}).get("frac"),
                                "matrix_pipe_frac": ((blk["roofline"] or {}).get("matrix_pipe") or {}).get("frac")}}
        if world == 1 and not args.no_secondary:
            sec = {}
            if T == 16:
                # BASELINE configs[4]: 32-frame 224^2 clips, head window [4,7,7] (stride_mod_layers="none"):
                # in the headline arithmetic, and in the configuration's stated form (bf16 activation storage)
                for key, m, sb, ss in (("s32", args.math, min(16, B), 1), ("s32_bf16act", "bf16act", min(16, B), 2)):
                    sec[key] = i3d_block(m, 32, sb, ss, 1,
                                         f"I3D mask search + Grad-CAM on 32-frame clips [{sb},3,32,224,224] per step, "
                                         f"stride_mod_layers='none' (BASELINE configs[4]"
                                         f"{'; bf16 activation / gradient storage, fp32 accumulate' if m == 'bf16act' else ''})")
            for key, m, sb, ss in (("bf16x3", "bf16x3", B, 2), ("fp32_mfma", "fp32", min(16, B), 3)):
                if m != args.math:
                    sec[key] = i3d_block(m, T, sb, ss, 1, f"same search, [{sb},3,{T},224,224] per step, {MODES[m][3]}")
            sec["convlstm"] = convlstm_block(torch, dev)
            out["secondary"] = sec
            for key, b_ in sec.items():
                r = b_.get("roofline") or {}
                summary[key] = {"clips_per_s": b_["value"], "roofline_frac": r.get("frac"), "bound": r.get("bound"),
                                "math": b_["config"].get("math", "f32")}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.iters, args.cpu_sample_iters, lam1, lam2, host_cores())
            summary["cpu_baseline"] = {"clips_per_s": round(out["cpu_baseline"]["value"], 5),
                                       "cores": out["cpu_baseline"]["cores"], "kind": out["cpu_baseline"]["kind"]}
        out["summary"] = summary      # LAST key: the figures of every block inside the tail of the line
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
