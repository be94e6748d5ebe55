"""fea_amd: MI355X-native (gfx950) implementation of the erichou1/fea UNet hot path.

    from fea_amd import FloorPlanSegmentationModel      # drop-in for data_model/models.py
"""
from .models import FloorPlanSegmentationModel  # noqa: F401
from . import functional  # noqa: F401
from .inference import Inference, TiledSegmenter  # noqa: F401

__all__ = ["FloorPlanSegmentationModel", "functional", "Inference", "TiledSegmenter"]
