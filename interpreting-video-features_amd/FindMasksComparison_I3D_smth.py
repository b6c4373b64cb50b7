"""Drop-in for video_features_pytorch/FindMasksComparison_I3D_smth.py on the MI355X.

    python FindMasksComparison_I3D_smth.py -c configs/config_i3d_smth.py --msl "" \
        --checkpoint ckpt.pth.tar --subDir run0 [--synthetic 16]

`find_masks` keeps the reference's positional signature (smth:125-126).
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import ivf_find_masks  # noqa: E402
import utils  # noqa: E402

RESIZE_SIZE_WIDTH = 224
RESIZE_SIZE_HEIGHT = 224
_state = {"sub_dir": "run0"}


def find_masks(dat_loader, model, hyper_params, lam1, lam2, N, maskType="gradient", temporalMaskType="freeze",
               classOI=None, verbose=True, maxMaskLength=None, doGradCam=False, runTempMask=True):
    """smth:125-315.  maskType is accepted for compatibility: the reference hard-codes
    mode="central" (smth:190)."""
    return ivf_find_masks.find_masks_impl(
        dat_loader, model, hyper_params, lam1, lam2, N, temporalMaskType, classOI, verbose, doGradCam,
        runTempMask, flavour="smth", sub_dir=_state["sub_dir"],
        gradcam_size=(RESIZE_SIZE_HEIGHT, RESIZE_SIZE_WIDTH))


def main(argv=None):
    args = utils.load_args(argv)
    config = utils.load_module(args.config).config
    cnn_def = importlib.import_module(config['conv_model'])
    device, device_ids = utils.setup_cuda_devices(args)
    torch.cuda.set_device(device)
    print(" > Using device: {}".format(device.type))
    print(" > Active GPU ids: {}".format(device_ids))
    _state["sub_dir"] = args.subDir
    msl = args.mod_stride_layers if args.mod_stride_layers is not None else config.get('stride_mod_layers', "")
    model = cnn_def.Model(config['num_classes'], last_stride=1, stride_mod_layers=msl, softMax=1).to(device)  # smth:55-58
    ivf_find_masks.load_checkpoint_into(model, args.checkpoint)
    lam1 = args.lam1 if args.lam1 is not None else 0.01            # smth:106-113
    lam2 = args.lam2 if args.lam2 is not None else 0.02
    N = args.optIter if args.optIter is not None else 300           # smth:116-119
    if args.synthetic:
        loader = ivf_find_masks.SyntheticLoader(args.synthetic, config['batch_size'],
                                                (3, config['clip_size'], 224, 224), config['num_classes'])
    else:
        import ivf_ingest
        loader = ivf_ingest.JpegFolderLoader(config['data_folder'] + "/validation/", clip_size=config['clip_size'],
                                             batch_size=config['batch_size'], layout="smth",
                                             drop_last=True, device=device)     # val_loader, smth:71-77
    config.setdefault("gradCamType", args.gradCamType)
    find_masks(loader, model, config, lam1, lam2, N, "central", config.get("maskPerturbType", "freeze"),
               classOI=args.subsetFile, doGradCam=True, runTempMask=True)


if __name__ == '__main__':
    main()
