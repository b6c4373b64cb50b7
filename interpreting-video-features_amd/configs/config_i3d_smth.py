# Config dict surface of the reference (configs/config_i3d_smth.py) -- keys read on the
# saliency path, plus the two the drivers read but the reference configs omit.
config = {
    "model_name": "modelI3d",
    "input_mode": "jpg",
    "data_folder": "",
    "num_workers": 2,
    "num_classes": 174,
    "batch_size": 16,
    "clip_size": 16,
    "conv_model": "models.I3D_doubled",
    "input_spatial_size": 224,
    "shuffle": 0,
    "soft_max": 1,
    "last_stride": 1,
    "stride_mod_layers": "",
    "dropout": 0.5,
    "pretrained_model_path": "no_ckpt",
    "maskPerturbType": "freeze",
    "gradCamType": "guessed",
}
