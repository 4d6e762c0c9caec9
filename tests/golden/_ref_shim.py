"""Container-only harness: import the reference's own Python files, unmodified,
from /root/reference so that golden input/output vectors can be generated.

The reference cannot be imported as shipped (its package __init__ files import
detectron2 / fvcore / cv2 / co-tracker, none of which are installed and none of
which can be fetched offline).  This module registers
*name-only* stand-ins in sys.modules for the handful of third-party names the
path's files touch, and stub *packages* whose __path__ points at the reference
directories so the reference's __init__.py files are skipped.  Nothing from the
reference is copied: its files are executed where they lie.

This file is test infrastructure.  It never runs on the GPU box
(/root/reference does not exist there) and nothing in s2d_amd imports it.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
MT = os.path.join(REF, "model_training")
KM = os.path.join(REF, "keymask_ident")


def available():
    return os.path.isdir(MT)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


class _Registry:
    def __init__(self, name):
        self._name = name
        self._map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._map[o.__name__] = o
                return o
            return deco
        self._map[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._map[name]


def _configurable(init_func=None, *, from_config=None):
    # identity: goldens construct modules with explicit keyword arguments
    if init_func is not None:
        return init_func
    return lambda f: f


class _Conv2d(nn.Conv2d):
    """detectron2.layers.Conv2d semantics: conv -> norm -> activation."""

    def __init__(self, *args, **kwargs):
        norm = kwargs.pop("norm", None)
        activation = kwargs.pop("activation", None)
        super().__init__(*args, **kwargs)
        self.norm = norm
        self.activation = activation

    def forward(self, x):
        x = F.conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


class _ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


def _get_norm(norm, out_channels):
    if norm is None or norm == "":
        return None
    assert norm == "GN", norm
    return nn.GroupNorm(32, out_channels)


def _c2_xavier_fill(module):
    nn.init.kaiming_uniform_(module.weight, a=1)
    if module.bias is not None:
        nn.init.constant_(module.bias, 0)


def _point_sample(input, point_coords, **kwargs):
    add_dim = False
    if point_coords.dim() == 3:
        add_dim = True
        point_coords = point_coords.unsqueeze(2)
    output = F.grid_sample(input, 2.0 * point_coords - 1.0, **kwargs)
    if add_dim:
        output = output.squeeze(3)
    return output


_installed = False


def install():
    global _installed
    if _installed:
        return
    _installed = True
    d2 = _mod("detectron2")
    _mod("detectron2.config", configurable=_configurable)
    _mod("detectron2.layers", Conv2d=_Conv2d, ShapeSpec=_ShapeSpec, get_norm=_get_norm,
         cat=lambda ts, dim=0: torch.cat(ts, dim=dim),
         shapes_to_tensor=lambda x, device=None: torch.as_tensor(x, device=device))
    _mod("detectron2.utils")
    _mod("detectron2.utils.registry", Registry=_Registry)
    _mod("detectron2.utils.comm", get_world_size=lambda: 1)
    _mod("detectron2.utils.memory", retry_if_cuda_oom=lambda f: f)
    _mod("detectron2.data", MetadataCatalog=None)
    _mod("detectron2.modeling", SEM_SEG_HEADS_REGISTRY=_Registry("SEM_SEG_HEADS"),
         META_ARCH_REGISTRY=_Registry("META_ARCH"), build_backbone=None, build_sem_seg_head=None)
    _mod("detectron2.modeling.backbone", Backbone=nn.Module)
    _mod("detectron2.modeling.postprocessing", sem_seg_postprocess=None)
    _mod("detectron2.structures", BitMasks=object, Boxes=object, ImageList=object, Instances=object)
    _mod("detectron2.projects")
    _mod("detectron2.projects.point_rend")
    _mod("detectron2.projects.point_rend.point_features", point_sample=_point_sample)
    fv = _mod("fvcore")
    fvnn = _mod("fvcore.nn")
    wi = _mod("fvcore.nn.weight_init", c2_xavier_fill=_c2_xavier_fill)
    fvnn.weight_init = wi
    fv.nn = fvnn
    # empty native module: forces MSDeformAttn.forward's bare `except:` onto the
    # reference's own pure-torch core (ms_deform_attn.py:116-121)
    _mod("MultiScaleDeformableAttention")
    # stub packages: skip the reference's __init__.py files
    _pkg("mask2former", os.path.join(MT, "mask2former"))
    _pkg("mask2former.modeling", os.path.join(MT, "mask2former", "modeling"))
    _pkg("mask2former.modeling.transformer_decoder",
         os.path.join(MT, "mask2former", "modeling", "transformer_decoder"))
    _pkg("mask2former.modeling.pixel_decoder", os.path.join(MT, "mask2former", "modeling", "pixel_decoder"))
    _pkg("mask2former.modeling.pixel_decoder.ops",
         os.path.join(MT, "mask2former", "modeling", "pixel_decoder", "ops"))
    _pkg("mask2former_video", os.path.join(MT, "mask2former_video"))
    _pkg("mask2former_video.modeling", os.path.join(MT, "mask2former_video", "modeling"))
    _pkg("mask2former_video.modeling.transformer_decoder",
         os.path.join(MT, "mask2former_video", "modeling", "transformer_decoder"))
    _pkg("mask2former_video.utils", os.path.join(MT, "mask2former_video", "utils"))
    _mod("mask2former_video.utils.debugging")  # visualisation helpers only (PIL / disk writes)
    _mod("mask2former_video.utils.memory", retry_if_cuda_oom=lambda f: f)
    _pkg("mask2former_video.data_video", os.path.join(MT, "mask2former_video", "data_video"))
    _mod("mask2former_video.data_video.dataset_mapper", apply_transformation_frame_by_frame=None,
         apply_transformslist_frame_by_frame=None)
    # keymask third parties
    _mod("cv2")
    _mod("cotracker")
    _mod("cotracker.predictor", CoTrackerPredictor=None)
    _mod("cotracker.utils")
    _mod("cotracker.utils.visualizer", Visualizer=None, read_video_from_path=None)
    if KM not in sys.path:
        sys.path.insert(0, KM)


def ref(name):
    """import a reference module by dotted name (after install())."""
    install()
    return importlib.import_module(name)
