"""
atlite_amd - MI355X (gfx950) implementation of PyPSA/atlite's convert_and_aggregate hot path.

HIP kernels + C ABI: ``atlite_amd/csrc`` -> ``atlite_amd/lib/libatlite_hip.so``
(``include/atlite_hip.h``).  Host-side mirror of the reference interface:
``atlite_amd.convert`` (``convert_and_aggregate``, ``pv``, ``wind``, ``heat_demand``,
``runoff``), ``atlite_amd.Cutout``, ``atlite_amd.Dataset``.
"""

from .cutout import Cutout
from .labeled import Dataset, LabeledArray
from .gis import indicatormatrix_of_grid as compute_indicatormatrix  # the reference's (orig, dest, orig_crs, dest_crs)
from .multigpu import set_devices
from .resource import solarpanels, windturbines

__version__ = "0.1.0"
__all__ = ["Cutout", "Dataset", "LabeledArray", "set_devices", "compute_indicatormatrix", "solarpanels", "windturbines"]
