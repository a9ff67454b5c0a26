"""Seeded planer-IR generators for the benchmark configs (no onnx needed)."""
from . import customnet, resnet18, yolov3
from .builder import GraphBuilder, blob_sha256, save_model

__all__ = ["customnet", "resnet18", "yolov3", "GraphBuilder", "blob_sha256",
           "save_model"]
