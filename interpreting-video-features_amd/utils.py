"""Flag / config surface of video_features_pytorch/utils.py that the saliency
drivers read: same flag names and config-module loader."""
import argparse
import importlib.util
import os

import torch


def load_args(argv=None):
    """utils.py:12-91 flag names (those read on the saliency path) with the same
    defaults; unknown flags are tolerated so reference command lines keep working."""
    p = argparse.ArgumentParser(description='MI355X temporal-mask / Grad-CAM saliency')
    p.add_argument('--config', '-c', help='python config module path holding `config = {...}`')
    p.add_argument('--eval_only', '-e', action='store_true')
    p.add_argument('--resume', '-r', action='store_true')
    p.add_argument('--gpus', '-g', default="0", help='GPU ids (one process per GPU; first id is used)')
    p.add_argument('--use_cuda', action='store_true')
    p.add_argument('--checkpoint', default="", help='checkpoint with state_dict (module.-prefixed or not)')
    p.add_argument('--subDir', default="run0")
    p.add_argument('--lam1', type=float, default=None)
    p.add_argument('--lam2', type=float, default=None)
    p.add_argument('--optIter', type=int, default=None)
    p.add_argument('--mod_stride_layers', '--msl', default=None,
                   help='comma separated endpoints whose temporal stride becomes last_stride; pass "" for none')
    p.add_argument('--dropout', type=float, default=0.5)
    p.add_argument('--gradCamType', default="guessed")
    p.add_argument('--subsetFile', default=None, help='csv of class -> clip ids (classOI)')
    p.add_argument('--synthetic', type=int, default=0, help='run on N synthetic clips instead of a dataset')
    args, _ = p.parse_known_args(argv)
    return args


def load_module(path):
    """utils.py:115-122."""
    spec = importlib.util.spec_from_file_location(os.path.splitext(os.path.basename(path))[0], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def setup_cuda_devices(args):
    """utils.py:134-139: (device, device_ids).  One process drives one GPU here."""
    ids = [int(i) for i in str(args.gpus).split(',') if i != ""] or [0]
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the saliency path runs on the MI355X only")
    return torch.device("cuda", ids[0]), ids


def remove_module_from_checkpoint_state_dict(state_dict):
    """utils.py:94-104."""
    return {(k[7:] if k.startswith('module.') else k): v for k, v in state_dict.items()}
