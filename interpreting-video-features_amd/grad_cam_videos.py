"""Drop-in for video_features_pytorch/grad_cam_videos.py (+ the names it re-exports
from the vendored pytorch-grad-cam/grad-cam.py via `from grad_cam import *`).

`GradCamVideo(model, ['Mixed_5c'], ...)(clip, index)` returns the same
`(cam_vid float32 [T,H,W], output [1,K])` pair, computed by the HIP plan: forward
to Mixed_5c, head backward from the (post-softmax) class score, channel-mean
weights, weighted sum + ReLU, bilinear resize, repeat, min/max normalise
(csrc/pool_head.hip; reference grad_cam_videos.py:64-142).
"""
import numpy as np
import torch

import ivf_lib as L


def _unwrap(model):
    return model.module if hasattr(model, "module") and not hasattr(model, "_engine_for") else model


def _one_target(target_layers):
    """Any ONE endpoint of the model may be the target layer.  With several, the reference pairs
    `features[-1]` (the LAST target in forward order) with `gradients[-1]` (the hook that fired last in the
    backward pass, i.e. the FIRST target): activations and gradients of different layers -- refused."""
    if len(target_layers) != 1:
        raise L.IvfError("GradCamVideo takes exactly one target layer (the reference mixes the activations of "
                         "the last target with the gradients of the first when given several)")
    return target_layers[0]


class FeatureExtractor():
    """pytorch-grad-cam/grad-cam.py:11-54: activations of the target layers and the
    output of the last feature module.  Gradients are produced by the HIP head
    backward (no autograd hooks)."""

    def __init__(self, model, target_layers, archType="I3D"):
        self.model = _unwrap(model)
        self.archType = archType
        self.target_layers = target_layers
        self.gradients = []

    def save_gradient(self, grad):
        self.gradients.append(grad)

    def __call__(self, x):
        if self.archType != "I3D":
            raise L.IvfError("FeatureExtractor: only archType 'I3D' is built; the reference's CLSTM branch "
                             "refers to attributes CLSTM_4.Model does not have (SURVEY.md F5)")
        _one_target(self.target_layers)
        self.gradients = []
        eng = self.model._engine_for(x)
        eng.forward(x)
        acts = [eng.endpoint(t, x.shape[0]) for t in self.target_layers]
        return acts, eng.endpoint('Mixed_5c', x.shape[0])


class ModelOutputs():
    """pytorch-grad-cam/grad-cam.py:56-71 (2-D image models) -- name kept for import
    compatibility; the 2-D VGG demo path is outside the video saliency path."""

    def __init__(self, model, target_layers):
        self.model = model
        self.target_layers = target_layers

    def get_gradients(self):
        return self.feature_extractor.gradients

    def __call__(self, x):
        raise L.IvfError("ModelOutputs (2-D pytorch-grad-cam) is not part of the video path; use ModelOutputsVideo")


class ModelOutputsVideo(ModelOutputs):
    """grad_cam_videos.py:13-43."""

    def __init__(self, model, target_layers, archType):
        self.model = _unwrap(model)
        self.archType = archType
        self.feature_extractor = FeatureExtractor(self.model, target_layers, archType)

    def __call__(self, x):
        target_activations, feat = self.feature_extractor(x)
        eng = self.model._engine_for(x)
        output = eng.forward(x)          # head of grad_cam_videos.py:30-41 (avg-pool, logits, softmax)
        return target_activations, output


class GradCam:
    """pytorch-grad-cam/grad-cam.py:96-145 (2-D) -- importable, not on the video path."""

    def __init__(self, model, target_layer_names, use_cuda):
        self.model = model
        self.cuda = use_cuda

    def forward(self, input):
        return self.model(input)

    def __call__(self, input, index=None):
        raise L.IvfError("GradCam (2-D) is not part of the video path; use GradCamVideo")


class GradCamVideo(GradCam):
    """grad_cam_videos.py:46-142."""

    def __init__(self, model, target_layer_names, class_dict, use_cuda,
                 input_spatial_size=224, normalizePerFrame=False, archType="I3D"):
        self.model = _unwrap(model)
        self.archType = archType
        self.cuda = use_cuda
        self.normalizePerFrame = normalizePerFrame
        # the reference calls len() on this (grad_cam_videos.py:54), which fails for the
        # int default; accept an int or a 1-/2-sequence (width, height)
        if isinstance(input_spatial_size, int):
            self.input_spatial_size = (input_spatial_size, input_spatial_size)
        elif len(input_spatial_size) == 1:
            self.input_spatial_size = (input_spatial_size[0], input_spatial_size[0])
        else:
            self.input_spatial_size = tuple(input_spatial_size)
        self.class_dict = class_dict
        if self.cuda:
            self.model = self.model.cuda()
        self.extractor = ModelOutputsVideo(self.model, target_layer_names, self.archType)

    def __call__(self, input, index=None):
        if self.archType != "I3D":
            raise L.IvfError("GradCamVideo: only archType 'I3D' is built (SURVEY.md F5)")
        layer = _one_target(self.extractor.feature_extractor.target_layers)
        x = input.cuda() if self.cuda else input
        L.require_gpu(x)
        if x.shape[0] != 1:
            raise L.IvfError("GradCamVideo takes one clip [1,C,T,H,W], as the reference does (smth:264)")
        eng = self.model._engine_for(x)
        target = None
        if index is not None:
            target = [int(index)]                           # grad_cam_videos.py:69-72
        width, height = self.input_spatial_size             # cv2 dsize order, grad_cam_videos.py:119-120
        cam, output = eng.gradcam(x, target, per_frame=bool(self.normalizePerFrame), out_hw=(height, width),
                                  layer=layer)
        cam_vid = cam[0].cpu().numpy().astype(np.float32)
        return cam_vid, output
