"""Shared body of the two drop-in drivers (FindMasksComparison_I3D_{smth,KTH}.py).

Reproduces the INTENDED behaviour of the reference's `find_masks` (neither
published driver runs as-is): per clip, baseline scores ->
init_mask('central') -> N Adam iterations -> reverse score -> Grad-CAM, result
dicts with the reference's keys, the two pickles and the per-clip
ClassScore{Freeze,Reverse}case<id>.txt files (smth:222-251, 272-277, 307-313;
KTH:273-303, 330-334, 372-378).  All clips of a loader batch are searched
together with per-clip masks.
"""
import csv
import os
import pickle

import numpy as np
import torch

import ivf_lib as L
import ivf_search


def _class_filter(classOI):
    """smth:147,173-175: csv whose columns are class ids and cells clip ids."""
    if classOI is None:
        return None
    table = {}
    with open(classOI, newline='') as f:
        rows = list(csv.reader(f))
    header = rows[0]
    for j, key in enumerate(header):
        vals = set()
        for r in rows[1:]:
            if j < len(r) and r[j] != '':
                try:
                    vals.add(int(float(r[j])))
                except ValueError:
                    pass
        table[str(key)] = vals
    return table


def _unwrap(model):
    return model.module if hasattr(model, "module") and not hasattr(model, "_engine_for") else model


def find_masks_impl(dat_loader, model, hyper_params, lam1, lam2, N, temporalMaskType="freeze", classOI=None,
                    verbose=True, doGradCam=False, runTempMask=True, flavour="smth", sub_dir="run0",
                    results_path="results/", gradcam_size=None, write_files=True, device=None, visualise=True):
    net = _unwrap(model)
    net.eval()                                                      # smth:145
    flt = _class_filter(classOI)
    masks, time_results, cam_results = [], [], []
    if write_files and not os.path.exists(results_path):
        os.makedirs(results_path)
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    for i, (sequence, label, video_id) in enumerate(dat_loader):
        if i % 50 == 0 and verbose:
            print("on idx: ", i)
        x = L.f32c(sequence.to(dev))                                # smth:158
        labels = torch.as_tensor(label).reshape(-1)
        ids = [v.item() if torch.is_tensor(v) else v for v in video_id] if not isinstance(video_id, str) else [video_id]
        keep = []
        for bi in range(x.shape[0]):                                # smth:166-175
            tc = str(int(labels[bi]))
            if flt is None or (tc in flt and int(ids[bi]) in flt[tc]):
                keep.append(bi)
        if not keep:
            continue
        xs = x[keep].contiguous()
        eng = net._engine_for(xs)
        search = ivf_search.MaskSearch(eng, lam1, lam2, N, temporalMaskType, threshold=0.9, lr=0.2,
                                       grad_cam_type=hyper_params.get("gradCamType", "guessed"),
                                       do_gradcam=doGradCam, run_temp_mask=runTempMask,
                                       normalize_per_frame=True, gradcam_size=gradcam_size)
        res = search.run(xs, labels[keep])
        host = {k: v.detach().cpu() for k, v in res.items() if k != "gradcam"}
        if doGradCam:
            host["gradcam"] = res["gradcam"].detach().cpu()
        for j, bi in enumerate(keep):
            true_class = int(labels[bi])
            pred = int(host["pred_class"][j])
            vid = ids[bi]
            gs = float(host["original_score_guess"][j])
            cs = float(host["original_score_true"][j])
            if runTempMask:
                tm = host["time_mask"][j].numpy()
                fz, rv = float(host["freeze_score"][j]), float(host["reverse_score"][j])
                if write_files:
                    gs_name = int(gs) if flavour == "smth" else gs     # smth:218 casts the score with int()
                    d = os.path.join("cam_saved_images", sub_dir, str(true_class),
                                     str(vid) + "g_" + str(pred) + "_gs%5.4f" % gs_name + "_cs%5.4f" % cs, "combined")
                    os.makedirs(d, exist_ok=True)
                    with open(os.path.join(d, "ClassScoreFreezecase" + str(vid) + ".txt"), "w+") as f:
                        f.write(str(fz))
                    with open(os.path.join(d, "ClassScoreReversecase" + str(vid) + ".txt"), "w+") as f:
                        f.write(str(rv))
                time_results.append({'true_class': true_class, 'pred_class': pred, 'video_id': vid,
                                     'time_mask': tm,
                                     'original_score_guess': int(gs) if flavour == "smth" else gs,
                                     'original_score_true': cs, 'freeze_score': fz, 'reverse_score': rv})
                tmask = res["time_mask"][j].clone()     # the clip's own [T] tensor, as the reference's time_mask
                if verbose:
                    print("resulting mask is: ", tmask)
            if doGradCam:
                cam_results.append({'true_class': true_class, 'pred_class': pred,
                                    'video_id': int(vid) if flavour == "smth" else vid,
                                    'GCHeatMap': host["gradcam"][j].numpy().astype(np.float32)})
            if runTempMask and write_files and visualise:
                # smth:296-303 / KTH:354-367: heat-map strips with the mask dot row for both perturbation
                # types (the dot row snaps a host copy: `tmask` keeps its sigmoid values, as the reference's
                # CUDA time_mask does), then for KTH the perturbed frames as PNGs of the SOFT mask
                import visualisation as viz
                if doGradCam:
                    for kind in ("freeze", "reverse"):
                        viz.create_image_arrays(xs, res["gradcam"][j], tmask, j, kind, d, str(vid), 0,
                                                xs.shape[4], xs.shape[3])
                if flavour != "smth":
                    import mask as _mask
                    viz.vizualize_results(xs[j], _mask.perturb_sequence(xs, tmask, temporalMaskType)[j], tmask,
                                          rootDir=d, case=str(vid), markImgs=True, iterTest=False)
            # smth:303 appends only when both Grad-CAM and the mask search ran; KTH:367 whenever the search ran
            if runTempMask and (doGradCam or flavour != "smth"):
                masks.append(tmask)
    if write_files:
        if flavour == "smth":                                       # smth:307-313
            tname = "allTimeMaskResults_" + sub_dir + "_" + str(classOI) + "_" + ".p"
            gname = "allGradCamResults_" + sub_dir + "_" + str(classOI) + "_" + ".p"
        else:                                                       # KTH:372-378
            tname = "I3d_KTH_allTimeMaskResults_original_" + sub_dir + ".p"
            gname = "I3d_KTH_allGradCamResults_original_" + sub_dir + ".p"
        with open(os.path.join(results_path, tname.replace(os.sep, "_")), "wb") as f:
            pickle.dump(time_results, f)
        with open(os.path.join(results_path, gname.replace(os.sep, "_")), "wb") as f:
            pickle.dump(cam_results, f)
    find_masks_impl.last_results = (time_results, cam_results)
    return masks


class SyntheticLoader:
    """Stand-in for ImLoader/KTHImLoader batches (data_loader_jpg.py:23-41): yields
    (sequence [B,3,T,H,W] float 0..255, label [B], video_id list) from ivf_recipe."""

    def __init__(self, n_clips, batch_size, shape, num_classes, first_id=0):
        self.n, self.bs, self.shape, self.k, self.first = n_clips, batch_size, shape, num_classes, first_id

    def __iter__(self):
        import ivf_recipe as R
        for s in range(0, self.n - self.n % self.bs if self.n >= self.bs else self.n, self.bs):
            ids = list(range(self.first + s, self.first + min(s + self.bs, self.n)))
            seq = torch.from_numpy(np.stack([R.clip(c, *self.shape) for c in ids]))
            yield seq, torch.tensor([R.label(c, self.k) for c in ids]), [str(c) for c in ids]


def load_checkpoint_into(model, checkpoint_path):
    """smth:89-103: a missing checkpoint is a printed warning, not an error."""
    if checkpoint_path and os.path.isfile(checkpoint_path):
        print(" > Loading checkpoint '{}'".format(checkpoint_path))
        ck = torch.load(checkpoint_path, map_location="cpu")
        _unwrap(model).load_state_dict(ck['state_dict'] if 'state_dict' in ck else ck)
        print(" > Loaded checkpoint '{}' (epoch {})".format(checkpoint_path, ck.get('epoch', '?')))
        return True
    print(" !#! No checkpoint found at '{}'".format(checkpoint_path))
    return False
