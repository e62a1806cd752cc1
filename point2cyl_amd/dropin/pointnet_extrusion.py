"""Drop-in for the reference's models/pointnet_extrusion.py (importlib.import_module('pointnet_extrusion').backbone)."""
from point2cyl_amd.backbone import backbone  # noqa: F401
