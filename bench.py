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

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "interpreting-video-features_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md: dense MFMA peaks
PEAK_TFLOPS = {"fp32": 157.3, "bf16x3": 2500.0}
PEAK_NOTE = {"fp32": "fp32 MFMA (v_mfma_f32_32x32x2_f32)",
             "bf16x3": "bf16 dense MFMA (v_mfma_f32_32x32x16_bf16); every algorithmic FLOP costs 3 MFMA FLOPs "
                       "in the split-bf16 mode, so the fp32-equivalent ceiling is 833 TFLOP/s"}
_HALO = ["4,192,32,96,1", "4,128,64,64,1", "4,128,32,64,1", "4,96,32,96,1", "4,64,32,64,1", "4,64,64,64,2",
         "4,32,32,32,1", "4,32,64,32,2", "2,192,32,96,1", "2,128,32,64,1", "2,96,32,96,1", "2,64,32,64,1",
         "2,32,32,32,1", "2,64,64,64,2", "2,128,64,64,1", "4,64,32,64,2", "4,96,32,96,2", "4,32,32,32,2",
         "2,192,32,96,1,16", "2,128,32,64,1,16", "4,96,32,96,1,16", "4,64,32,64,1,16", "2,96,32,96,1,16",
         "4,192,32,96,1,32,4,14", "4,128,32,64,1,32,4,14", "4,96,32,96,1,32,4,14", "4,64,32,64,1,32,4,14",
         "4,32,32,32,1,32,4,14", "4,64,32,32,1,32,4,14", "4,32,32,32,2,32,4,14", "4,128,32,128,1,32,4,14"]
_IGEMM = ["128,128,2,2", "128,64,4,1", "128,32,4,1"]
NCLASS = 48


def variant_name(v):
    """kernel template instance behind a profiler class id (include/ivf_hip.h)"""
    if v == 47:
        return "conv3d_pix4_kernel"
    if v >= 16:
        name = _HALO[v - 16]
        if name.count(',') == 4:
            name += ',32'
        return f"conv3d_halo_kernel<{name if name.count(',') == 7 else name + ',8,8'}>"
    if v in (10, 11):
        return f"conv3d_igemm_bf16x3_kernel<{'64,64,2,2' if v == 10 else '64,128,2,2'}>"
    if v >= 7:
        return f"conv3d_igemm_bf16x3_kernel<{['128,256,4,1', '128,192,4,1', '128,160,4,1'][v - 7]}>"
    if v >= 4:
        return f"conv3d_igemm_bf16x3_kernel<{_IGEMM[v - 4]}>"
    return f"conv3d_igemm_kernel<{_IGEMM[v - 1]}>"


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
    return min(n, 16)     # a one-GPU box is given a 16-CPU share


def cpu_baseline(n_iter_total, sample_iters, lam1, lam2, threads):
    """The CPU oracle (torch fp32 restatement of the reference path, oracle/) on the
    host cores of this box: one clip, init_mask + `sample_iters` iterations + reverse
    score + Grad-CAM; the iteration cost is extrapolated linearly to n_iter_total."""
    import ivf_recipe as R
    from oracle import gradcam_ref, i3d_ref, mask_ref
    torch.set_num_threads(threads)
    sd = R.to_torch(R.i3d_state_dict(num_classes=174))
    x = torch.from_numpy(R.clip(1000))[None]
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
    if t_init > 45:            # keep the CPU leg bounded on a slow host
        sample_iters = min(sample_iters, 2)
    t0 = time.perf_counter()
    res = mask_ref.search_clip(x, score_fn, lam1, lam2, sample_iters, init=tm)   # includes reverse score
    t_loop = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.no_grad():
        score_fn(mask_ref.perturb_sequence(x, res['mask'], 'reverse'))
    t_rev = time.perf_counter() - t0
    t0 = time.perf_counter()
    gradcam_ref.gradcam_i3d(x, sd, None)
    t_gc = time.perf_counter() - t0
    per_iter = (t_loop - t_rev) / sample_iters
    total = t_init + per_iter * n_iter_total + t_rev + t_gc
    return dict(value=1.0 / total, unit="clips/s", cores=threads, kind="port",
                sample=(f"1 clip [1,3,16,224,224]: init_mask {t_init:.1f}s + {sample_iters} of {n_iter_total} "
                        f"iterations ({per_iter:.2f}s each, extrapolated linearly) + reverse {t_rev:.1f}s + "
                        f"Grad-CAM {t_gc:.1f}s"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128, help="clips per GPU per step (searched together)")
    ap.add_argument("--iters", type=int, default=300, help="Adam iterations per search (reference N=300)")
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-iters", type=int, default=4)
    ap.add_argument("--math", choices=["fp32", "bf16x3"], default=None,
                    help="arithmetic of the Unit3D convolutions (default: ivf_engine.DEFAULT_MATH)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL on ROCm

    import ivf_engine
    import ivf_lib as L
    import ivf_recipe as R
    import ivf_search
    import ivf_shard

    lam1, lam2 = 0.01, 0.02   # FindMasksComparison_I3D_smth.py:106-113
    T, B = args.frames, args.batch
    eng = ivf_engine.I3DEngine(174, (3, T, 224, 224), max_batch=B, softmax=True,
                               stride_mod_layers="" if T == 16 else "none", device=dev, math=args.math)
    eng.load_state_dict(R.i3d_state_dict(num_classes=174), autotune=(rank == 0))
    if world > 1:
        # every rank must run the SAME kernel variant per layer (bit-identical per-clip results):
        # rank 0 tunes, one small broadcast installs its choice everywhere
        tune = torch.tensor(eng.get_tuning(), dtype=torch.int32, device=dev)
        dist.broadcast(tune, src=0)
        eng.set_tuning(tune.cpu().tolist())
    searcher = ivf_search.MaskSearch(eng, lam1, lam2, args.iters, "freeze", grad_cam_type="guessed",
                                     do_gradcam=True)
    # synthetic clips, resident in HBM before the timed region; shard: clip_id % world == rank
    n_steps_total = args.warmup + args.steps
    clip_ids = [[(s * B + i) * world + rank for i in range(B)] for s in range(n_steps_total)]
    uniq = sorted({c % 16 for ids in clip_ids for c in ids})      # 16 distinct synthetic clips, reused
    bank = {c: torch.from_numpy(R.clip(c, 3, T, 224, 224)).to(dev) for c in uniq}
    batches = [torch.stack([bank[c % 16] for c in ids]) for ids in clip_ids]
    labels = [[R.label(c, 174) for c in ids] for ids in clip_ids]

    def step(i):
        res = searcher.run(batches[i], labels[i])
        rec = ivf_search.pack_records(clip_ids[i], res, T)
        return ivf_shard.gather_records(rec), res      # one RCCL all_gather when world > 1

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    # sample conv launches on every 16th iteration with HIP events on the launch stream
    sample_every = 16
    L.check(L.lib().ivf_profile_enable(sample_every, 200000))
    fence()
    t0 = time.perf_counter()
    last = None
    for i in range(args.warmup, n_steps_total):
        last = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms = (ctypes.c_double * NCLASS)()
    launches = (ctypes.c_longlong * NCLASS)()
    flops = (ctypes.c_double * NCLASS)()
    L.check(L.lib().ivf_profile_collect(ms, launches, flops))
    L.check(L.lib().ivf_profile_disable())

    if rank == 0:
        clips = world * B * args.steps
        value = clips / elapsed
        math = eng.math
        peak = PEAK_TFLOPS[math]
        dom = int(np.argmax([ms[v] for v in range(NCLASS)]))
        roofline = None
        if launches[dom] > 0:
            avg_ms = ms[dom] / launches[dom]
            achieved = (flops[dom] / launches[dom]) / (avg_ms * 1e-3) / 1e12
            tot_ms = sum(ms[v] for v in range(NCLASS))
            tot_fl = sum(flops[v] for v in range(NCLASS))
            shares = {variant_name(v): round(ms[v] / tot_ms, 3) for v in range(NCLASS) if launches[v] > 0}
            # HBM bytes per launch of that kernel from the committed PMC passes (profiles/), if the
            # same kernel variant was measured there at this batch size; else null
            traffic = None
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
                ent = pm["kernels"].get(variant_name(dom))
                if ent and pm.get("batch") == B:
                    traffic = ent["hbm_bytes_per_launch"]
            except (OSError, ValueError, KeyError):
                pass
            roofline = {
                "bound": "mfma", "achieved": round(achieved, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_source": "profiles/r01_pmc_hbm_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                  "passes, FETCH doubled per the gfx950 rule)" if traffic else None,
                "kernel": variant_name(dom), "avg_launch_ms": round(avg_ms, 4),
                "sampled_launches": int(launches[dom]),
                "algorithmic_gflop_per_launch": round(flops[dom] / launches[dom] / 1e9, 3),
                "mfma_passes_per_algorithmic_flop": 3 if math == "bf16x3" else 1,
                "all_conv_kernels": {"achieved": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                                     "frac": round(tot_fl / (tot_ms * 1e-3) / 1e12 / peak, 4),
                                     "share_of_sampled_conv_time": shares},
                "peak_dtype": PEAK_NOTE[math],
            }
        out = {
            "metric": "clips/sec full mask-search (I3D, 16f, 300 iters)",
            "value": round(value, 4), "unit": "clips/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if eng.math == "fp32" else "bf16x3 (split-bf16 MFMA, f32 accumulate, f32 storage)",
            "data": "synthetic",
            "config": {"workload": f"I3D perturbation mask search, {args.iters} iters, synthetic clips "
                                   f"[{B},3,{T},224,224] per GPU per step (BASELINE configs[1]); init_mask + "
                                   f"search + reverse score + Grad-CAM",
                       "clips_per_gpu_per_step": B, "iters": args.iters, "frames": T,
                       "lam1": lam1, "lam2": lam2, "sharding": f"clip_id % {world}, all_gather of records"},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.iters, args.cpu_sample_iters, lam1, lam2, host_cores())
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
