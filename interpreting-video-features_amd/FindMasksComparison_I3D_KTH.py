"""Drop-in for video_features_pytorch/FindMasksComparison_I3D_KTH.py on the MI355X
(I3D-KTH or CLSTM_4 backbone, KTH:50-58).  `find_masks` keeps the KTH arity with
`ita` (KTH:126-127)."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import ivf_find_masks  # noqa: E402
import utils  # noqa: E402

RESIZE_SIZE_WIDTH = 160
RESIZE_SIZE_HEIGHT = 120
_state = {"sub_dir": "run0"}


def find_masks(dat_loader, model, config, lam1, lam2, N, ita=1, maskType="gradient", temporalMaskType="freeze",
               classOI=None, verbose=True, maxMaskLength=None, doGradCam=False, runTempMask=True):
    """KTH:126-380."""
    return ivf_find_masks.find_masks_impl(
        dat_loader, model, config, lam1, lam2, N, temporalMaskType, classOI, verbose, doGradCam, runTempMask,
        flavour="kth", sub_dir=_state["sub_dir"], gradcam_size=(RESIZE_SIZE_HEIGHT, RESIZE_SIZE_WIDTH))


def build_model(config, args, device):
    cnn_def = importlib.import_module(config['conv_model'])
    if config['conv_model'].endswith("CLSTM_4"):                     # KTH:53-58
        return cnn_def.Model(num_classes=config['num_classes'], nb_lstm_units=config['clstm_hidden'],
                             channels=3, conv_kernel_size=(5, 5), lstm_layers=config['clstm_layers'],
                             step=config['clip_size'], image_size=(160, 120), conv_stride=config['conv_stride'],
                             effective_step=[7, 15, 23, 31], dropout=args.dropout).to(device)
    msl = args.mod_stride_layers if args.mod_stride_layers is not None else config.get('stride_mod_layers', "")
    return cnn_def.Model(config['num_classes'], last_stride=1, stride_mod_layers=msl,
                         finalTimeLength=config.get('final_temp_time', 4), softMax=1).to(device)   # KTH:50-52


def main(argv=None):
    args = utils.load_args(argv)
    config = utils.load_module(args.config).config
    device, device_ids = utils.setup_cuda_devices(args)
    torch.cuda.set_device(device)
    _state["sub_dir"] = args.subDir
    model = build_model(config, args, device)
    ivf_find_masks.load_checkpoint_into(model, args.checkpoint)
    lam1 = args.lam1 if args.lam1 is not None else 0.02            # KTH:105-112
    lam2 = args.lam2 if args.lam2 is not None else 0.04
    N = args.optIter if args.optIter is not None else 100           # KTH:115-118
    if args.synthetic:
        loader = ivf_find_masks.SyntheticLoader(args.synthetic, config['batch_size'],
                                                (3, config['clip_size'], 120, 160), config['num_classes'])
    else:
        import ivf_ingest
        loader = ivf_ingest.JpegFolderLoader(config['data_folder'] + "/test", clip_size=config['clip_size'],
                                             batch_size=config['batch_size'], layout="kth",
                                             drop_last=True, device=device)     # val_loader, KTH:73-78
    config.setdefault("gradCamType", args.gradCamType)
    find_masks(loader, model, config, lam1, lam2, N, 1, "central", config.get("maskPerturbType", "freeze"),
               classOI=None, doGradCam=config['conv_model'].endswith("CLSTM_4") is False, runTempMask=True)


if __name__ == '__main__':
    main()
